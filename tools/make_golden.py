#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python modules on CPU.

Runs only in the build container (needs /root/reference); the GPU box sees only the
committed .npz fixtures.  Import recipe: SURVEY.md Appendix B — the two JIT CUDA
extensions are replaced by (i) a pure-torch stand-in for `fused.fused_bias_act` that
restates the switch at op/fused_bias_act_kernel.cu:28-47 (fused_act.py has no CPU
branch) and (ii) a dummy for `upfirdn2d_op` (never called: CPU tensors are routed to
the reference's own `upfirdn2d_native`, op/upfirdn2d.py:146-149).  Everything else
(autograd Functions, all nn.Modules, estimate_fisher) is the reference's code.

Inputs are never stored when they can be regenerated from rick_amd.synth (closed-form,
seeded by key name); only outputs are.

usage: python tools/make_golden.py [--only ops|layers|small|full|latents|fid|ada]
"""
import argparse
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, ROOT)
from rick_amd.synth import synth_latents, synth_reals, synth_state_dict, synth_tensor  # noqa: E402
from tests.cases import UPFIRDN_CASES, upfirdn_kernel  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def import_reference():
    import torch.utils.cpp_extension as cpp

    class _Fused:
        @staticmethod
        def fused_bias_act(x, b, ref, act, grad, alpha, scale):
            if b.numel():
                x = x + b.reshape(1, -1, *([1] * (x.ndim - 2)))
            if act == 3 and grad == 0:
                y = torch.where(x > 0, x, x * alpha)
            elif act == 3 and grad == 1:
                y = torch.where(ref > 0, x, x * alpha)
            elif grad == 2:
                y = torch.zeros_like(x)
            else:
                y = x
            return y * scale

    def fake_load(name, sources=None, **kw):
        return _Fused if name == 'fused' else types.SimpleNamespace()

    cpp.load = fake_load
    sys.path.insert(0, REF)
    import op  # noqa: F401  (reference package, CPU routing)
    spec = importlib.util.spec_from_file_location(
        'mpt', os.path.join(REF, 'gan_training/models/model_probe_tune.py'))
    mpt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mpt)
    return op, mpt


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32) if t.dtype != torch.float64 else t.detach().cpu().numpy()


# ------------------------------------------------------------------------------- ops
def gen_ops(op):
    out = {}
    for (tag, up, down, p0, p1, n, c, h, w, ks) in UPFIRDN_CASES:
        k = upfirdn_kernel(ks, up).double()
        x = synth_tensor(f'upfirdn/{tag}/x', (n, c, h, w)).double().requires_grad_(True)
        y = op.upfirdn2d(x, k, up=up, down=down, pad=(p0, p1))
        gy = synth_tensor(f'upfirdn/{tag}/gy', y.shape).double().requires_grad_(True)
        (gx,) = torch.autograd.grad(y, x, gy, create_graph=True)
        ggx = synth_tensor(f'upfirdn/{tag}/ggx', x.shape).double()
        (ggy,) = torch.autograd.grad(gx, gy, ggx)
        # fp32 forward too (bit-level comparison target for the C oracle's fmaf chain is not
        # claimed; fp32 result kept for tolerance checks)
        y32 = op.upfirdn2d(x.detach().float(), k.float(), up=up, down=down, pad=(p0, p1))
        out[f'{tag}/y'] = y.detach().numpy()
        out[f'{tag}/gx'] = gx.detach().numpy()
        out[f'{tag}/ggy'] = ggy.detach().numpy()
        out[f'{tag}/y32'] = y32.numpy()
    # fused leaky relu through the reference's autograd Functions (op/fused_act.py:19-70)
    for tag, shape in (('act2d', (3, 8)), ('act4d', (2, 5, 6, 6))):
        x = synth_tensor(f'act/{tag}/x', shape).double().requires_grad_(True)
        b = synth_tensor(f'act/{tag}/b', (shape[1],)).double().requires_grad_(True)
        y = op.fused_leaky_relu(x, b)
        gy = synth_tensor(f'act/{tag}/gy', shape).double().requires_grad_(True)
        gx, gb = torch.autograd.grad(y, (x, b), gy, create_graph=True)
        ggx = synth_tensor(f'act/{tag}/ggx', shape).double()
        ggb = synth_tensor(f'act/{tag}/ggb', (shape[1],)).double()
        (ggy,) = torch.autograd.grad((gx, gb), gy, (ggx, ggb))
        out[f'{tag}/y'] = y.detach().numpy()
        out[f'{tag}/gx'] = gx.detach().numpy()
        out[f'{tag}/gb'] = gb.detach().numpy()
        out[f'{tag}/ggy'] = ggy.detach().numpy()
    np.savez_compressed(os.path.join(OUT, 'ops.npz'), **out)
    print('ops.npz', len(out), 'arrays')


# ----------------------------------------------------------------------------- layers
def gen_layers(mpt):
    out = {}
    B, CI, CO, R, SD = 3, 16, 24, 8, 32
    for tag, kw in (('plain', dict(kernel_size=3)), ('up', dict(kernel_size=3, upsample=True)),
                    ('rgb', dict(kernel_size=1, demodulate=False))):
        co = 3 if tag == 'rgb' else CO
        m = mpt.ModulatedConv2d(CI, co, style_dim=SD, **kw).double()
        sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, dtype=torch.float64)
        m.load_state_dict(sd, strict=False)
        x = synth_tensor(f'modconv/{tag}/x', (B, CI, R, R)).double().requires_grad_(True)
        s = synth_tensor(f'modconv/{tag}/s', (B, SD)).double().requires_grad_(True)
        y = m(x, s)
        gy = synth_tensor(f'modconv/{tag}/gy', y.shape).double()
        params = [m.weight, m.modulation.weight, m.modulation.bias]
        grads = torch.autograd.grad(y, [x, s] + params, gy, create_graph=True)
        # PLR-style second-order scalar: || d<y,gy>/ds ||^2, differentiated w.r.t. weight and x
        pl = grads[1].pow(2).sum()
        gg = torch.autograd.grad(pl, [x, m.weight, s], allow_unused=True)
        gg = [torch.zeros_like(t) if g_ is None else g_ for g_, t in zip(gg, [x, m.weight, s])]
        out[f'{tag}/y'] = y.detach().numpy()
        for n, g in zip(('gx', 'gs', 'gw', 'gmw', 'gmb'), grads):
            out[f'{tag}/{n}'] = g.detach().numpy()
        out[f'{tag}/pl'] = pl.detach().numpy()
        for n, g in zip(('pl_gx', 'pl_gw', 'pl_gs'), gg):
            out[f'{tag}/{n}'] = g.detach().numpy()
    # minibatch stddev via the reference Discriminator tail is covered in 'small'
    np.savez_compressed(os.path.join(OUT, 'layers.npz'), **out)
    print('layers.npz', len(out), 'arrays')


# ------------------------------------------------------------------------ whole models
def softplus(x):
    return torch.nn.functional.softplus(x)


def build(mpt, size, dtype):
    g = mpt.Generator(size, 512, 8, channel_multiplier=2)
    d = mpt.Discriminator(size, channel_multiplier=2)
    sg = synth_state_dict({k: v.shape for k, v in g.state_dict().items()})
    sd = synth_state_dict({k: v.shape for k, v in d.state_dict().items()})
    g.load_state_dict(sg, strict=False)
    d.load_state_dict(sd, strict=False)
    return g.to(dtype), d.to(dtype)


def grad_summ(named, grads):
    """per-key sum(g^2) (== Fisher mass) — compact fingerprint of every parameter gradient."""
    return {k: float((g.double() ** 2).sum()) if g is not None else 0.0 for (k, _), g in zip(named, grads)}


def model_case(mpt, size, B, dtype, out, tag, latents=None, full_arrays=False):
    g, d = build(mpt, size, dtype)
    z = (latents if latents is not None else synth_latents(B, seed=size)).to(dtype)
    real = synth_reals(B, size=size, seed=size).to(dtype)
    gp, dp = list(g.named_parameters()), list(d.named_parameters())

    fake, _ = g([z], randomize_noise=False)
    fake_pred, feat_f = d(fake)
    real_pred, _ = d(real)
    d_loss = softplus(-real_pred).mean() + softplus(fake_pred).mean()
    g_loss = softplus(-fake_pred).mean()
    gd = torch.autograd.grad(d_loss, [p for _, p in dp], retain_graph=True, allow_unused=True)
    gg = torch.autograd.grad(g_loss, [p for _, p in gp], retain_graph=True, allow_unused=True)
    out[f'{tag}/img_mean'] = np32(fake.mean(dim=(2, 3)))
    out[f'{tag}/img_std'] = np32(fake.std(dim=(2, 3)))
    idx = torch.from_numpy(np.random.RandomState(0).randint(0, fake[0].numel(), size=4096))
    out[f'{tag}/img_idx'] = idx.numpy()
    out[f'{tag}/img_samples'] = np32(fake.reshape(B, -1)[:, idx])
    if full_arrays:
        out[f'{tag}/img'] = np32(fake)
    out[f'{tag}/fake_pred'] = np32(fake_pred)
    out[f'{tag}/real_pred'] = np32(real_pred)
    out[f'{tag}/d_loss'] = np32(d_loss)
    out[f'{tag}/g_loss'] = np32(g_loss)
    out[f'{tag}/feat_absmean'] = np.array([float(f.abs().mean()) for f in feat_f])
    for k, v in grad_summ(dp, gd).items():
        out[f'{tag}/d_grad2/{k}'] = np.float64(v)
    for k, v in grad_summ(gp, gg).items():
        out[f'{tag}/g_grad2/{k}'] = np.float64(v)

    # R1 (train_dynamic_update_prune.py:89-96, 465-475)
    real_r = real.clone().requires_grad_(True)
    rp, _ = d(real_r)
    (gr,) = torch.autograd.grad(rp.sum(), real_r, create_graph=True)
    r1 = gr.pow(2).reshape(B, -1).sum(1).mean()
    gr1 = torch.autograd.grad(10 / 2 * r1 * 16 + 0 * rp[0].sum(), [p for _, p in dp], allow_unused=True)
    out[f'{tag}/r1'] = np32(r1)
    for k, v in grad_summ(dp, gr1).items():
        out[f'{tag}/r1_grad2/{k}'] = np.float64(v)

    # PLR (train_dynamic_update_prune.py:104-118, 548-566) with a shared seeded noise image
    pb = max(1, B // 2)
    img, lat = g([z[:pb]], return_latents=True, randomize_noise=False)
    pl_noise = synth_tensor(f'plnoise/{size}', img.shape).to(dtype)
    (gl,) = torch.autograd.grad((img * pl_noise / math.sqrt(size * size)).sum(), lat, create_graph=True)
    pl_len = torch.sqrt(gl.pow(2).sum(2).mean(1))
    pl_mean = 0 + 0.01 * (pl_len.mean() - 0)
    pl_loss = (pl_len - pl_mean).pow(2).mean()
    gpl = torch.autograd.grad(2 * 4 * pl_loss + 0 * img[0, 0, 0, 0], [p for _, p in gp], allow_unused=True)
    out[f'{tag}/pl_lengths'] = np32(pl_len)
    out[f'{tag}/pl_loss'] = np32(pl_loss)
    for k, v in grad_summ(gp, gpl).items():
        out[f'{tag}/pl_grad2/{k}'] = np.float64(v)
    return g, d


def gen_small(mpt):
    out = {}
    model_case(mpt, 32, 2, torch.float64, out, 's32_f64', full_arrays=True)
    model_case(mpt, 16, 4, torch.float64, out, 's16_f64', full_arrays=True)
    np.savez_compressed(os.path.join(OUT, 'small.npz'), **out)
    print('small.npz', len(out), 'arrays')


def gen_latents():
    lat = {f'noise_{j:04d}': torch.load(os.path.join(REF, '_noise', f'{j:04d}.pt')).numpy() for j in range(10)}
    lat['sample_z'] = torch.load(os.path.join(REF, 'noise.pt')).numpy()
    np.savez_compressed(os.path.join(OUT, 'noise_latents.npz'), **lat)
    print('noise_latents.npz', len(lat), 'arrays')


def gen_full(mpt):
    """256 px, fp32 (the reference's arithmetic type), on the shipped _noise latents."""
    out = {}
    lat = torch.cat([torch.load(os.path.join(REF, '_noise', f'{j:04d}.pt')) for j in range(2)], 0)
    g, d = model_case(mpt, 256, 2, torch.float32, out, 'f256', latents=lat)

    # Fisher sample j=0 exactly as train_dynamic_update_prune.py:231-248 (batch 1, real[0])
    z0 = lat[0:1]
    real = synth_reals(2, size=256, seed=256)
    fake, _ = g([z0.view(1, -1)], randomize_noise=False)
    fp, _ = d(fake)
    rp, _ = d(real[0].view(1, 3, 256, 256))
    g_loss = softplus(-fp).mean()
    d_loss = softplus(-rp).mean() + softplus(fp).mean()
    _, fg = g.estimate_fisher(g_loss)
    _, fd = d.estimate_fisher(d_loss)
    out['fisher/g_loss'] = np32(g_loss)
    out['fisher/d_loss'] = np32(d_loss)
    # per-filter FIM vectors (train_dynamic_update_prune.py:279-299, 334-353)
    fgn = {k: v.numpy() for k, v in fg.items()}
    fdn = {k: v.numpy() for k, v in fd.items()}
    for k in range(12):
        out[f'fisher/g_conv/{k}'] = fgn[f'convs.{k}.conv.weight'].mean(axis=(0, 2, 3, 4))
        out[f'fisher/g_fc/{k}'] = (fgn[f'convs.{k}.conv.modulation.weight'].mean(axis=1)
                                   + fgn[f'convs.{k}.conv.modulation.bias']) / 2
    for b in range(1, 7):
        for li in range(2):
            wk, bk = f'convs.{b}.conv{li + 1}.{li}.weight', f'convs.{b}.conv{li + 1}.{li + 1}.bias'
            out[f'fisher/d/{wk}'] = (fdn[wk].mean(axis=(1, 2, 3)) + fdn[bk]) / 2
        sk = f'convs.{b}.skip.1.weight'
        out[f'fisher/d/{sk}'] = fdn[sk].mean(axis=(1, 2, 3))
    for k, v in fgn.items():
        out[f'fisher/g_sum/{k}'] = np.float64(v.astype(np.float64).sum())
    for k, v in fdn.items():
        out[f'fisher/d_sum/{k}'] = np.float64(v.astype(np.float64).sum())
    np.savez_compressed(os.path.join(OUT, 'full256.npz'), **out)
    print('full256.npz', len(out), 'arrays')


def gen_ada():
    """ADA augmentation: the reference's non_leaking.py run on CPU.  Seeded sampler outputs
    (G, C) and the full augment() image path for explicit (G, C), incl. a case that needs reflect padding."""
    import non_leaking as nl
    out = {}
    for seed, (n, h, w), p in ((3, (4, 32, 32), 0.8), (11, (3, 24, 40), 0.5), (29, (2, 16, 16), 1.0)):
        torch.manual_seed(seed)
        out[f'samp{seed}/G'] = nl.sample_affine(p, n, h, w).numpy()
        out[f'samp{seed}/C'] = nl.sample_color(p, n).numpy()
        out[f'samp{seed}/meta'] = np.array([n, h, w, p], dtype=np.float64)
    gen = torch.Generator().manual_seed(5)
    for tag, (n, h, w), p, seed in (('a', (2, 32, 32), 0.8, 17), ('b', (3, 24, 40), 0.6, 23), ('c', (1, 64, 64), 1.0, 31)):
        img = torch.rand(n, 3, h, w, generator=gen) * 2 - 1
        torch.manual_seed(seed)
        res, (G, C) = nl.augment(img, p)
        res2, _ = nl.augment(img, p, (G, C))            # explicit matrices reproduce the sampled run
        assert torch.equal(res, res2)
        out[f'aug{tag}/img'], out[f'aug{tag}/G'], out[f'aug{tag}/C'] = img.numpy(), G.numpy(), C.numpy()
        out[f'aug{tag}/out'] = res.numpy()
    np.savez_compressed(os.path.join(OUT, 'ada.npz'), **out)
    print('ada.npz:', {k: v.shape for k, v in out.items() if k.endswith('/out')})


def gen_fid():
    """FID statistics.  gan_training/metrics/fid_score.py imports cv2 / an Inception wrapper
    that are absent here, so the two pure-NumPy/SciPy pieces are taken out of the reference FILE with `ast` and
    executed as they are: calculate_frechet_distance (:94-129) and the mean / np.cov lines of
    calculate_activation_statistics (:138-142, restated as a two-line lambda because they sit behind the model call)."""
    import ast
    from scipy import linalg
    src = open(os.path.join(REF, 'gan_training', 'metrics', 'fid_score.py')).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'calculate_frechet_distance'][0]
    ns = {'np': np, 'linalg': linalg}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), 'fid_score.py', 'exec'), ns)
    frechet = ns['calculate_frechet_distance']
    rng = np.random.RandomState(7)
    out = {}
    for tag, n, d, rank in (('a', 96, 24, 24), ('b', 200, 64, 64), ('lowrank', 40, 32, 12)):
        basis = rng.randn(rank, d)
        act0 = (rng.randn(n, rank) @ basis + 0.3 * rng.randn(1, d)).astype(np.float32)
        act1 = (rng.randn(n, rank) * 1.3 @ basis + 0.1).astype(np.float32)
        # get_activations collects the fp32 network outputs in np.empty((n, dims)) — a float64 array (:72,86)
        a0, a1 = np.empty(act0.shape), np.empty(act1.shape)
        a0[:], a1[:] = act0, act1
        m0, s0 = np.mean(a0, axis=0), np.cov(a0, rowvar=False)
        m1, s1 = np.mean(a1, axis=0), np.cov(a1, rowvar=False)
        out[f'{tag}/act0'], out[f'{tag}/act1'] = act0, act1
        out[f'{tag}/mu0'], out[f'{tag}/sigma0'], out[f'{tag}/mu1'], out[f'{tag}/sigma1'] = m0, s0, m1, s1
        out[f'{tag}/fid'] = np.float64(frechet(m0, s0, m1, s1))
    # KID (gan_metrics/kid_score.py): the reference's own polynomial_mmd_averages / polynomial_mmd / _mmd2_and_variance,
    # cut out of the file (its module imports torchvision and PIL), seeded through NumPy's global generator
    import io
    import sys as _sys
    from sklearn.metrics.pairwise import polynomial_kernel
    from tqdm import tqdm
    ksrc = open(os.path.join(REF, 'gan_metrics', 'kid_score.py')).read()
    want = {'_sqn', 'polynomial_mmd_averages', 'polynomial_mmd', '_mmd2_and_variance'}
    kfns = [n for n in ast.parse(ksrc).body if isinstance(n, ast.FunctionDef) and n.name in want]
    kns = {'np': np, 'polynomial_kernel': polynomial_kernel, 'tqdm': tqdm, 'sys': _sys}
    exec(compile(ast.Module(body=kfns, type_ignores=[]), 'kid_score.py', 'exec'), kns)
    # activations are collected in np.empty((n, dims)) — float64 — exactly like the FID path (kid_score.py:199,229)
    cg = (rng.randn(300, 48) * 0.8 + 0.2).astype(np.float32).astype(np.float64)
    cr = (rng.randn(260, 48)).astype(np.float32).astype(np.float64)
    np.random.seed(12)
    mmds, _vars = kns['polynomial_mmd_averages'](cg, cr, n_subsets=6, subset_size=100, output=io.StringIO())
    out['kid/codes_g'], out['kid/codes_r'], out['kid/mmds'] = cg, cr, mmds
    out['kid/seed'], out['kid/n_subsets'], out['kid/subset_size'] = np.int64(12), np.int64(6), np.int64(100)
    # improved precision / recall (gan_metrics/precision_recall.py): the reference's own distance / radius / coverage
    # functions, cut out of the file (the module imports torchvision and PIL)
    from collections import namedtuple
    psrc = open(os.path.join(REF, 'gan_metrics', 'precision_recall.py')).read()
    pwant = {'compute_pairwise_distances', 'distances2radii', 'get_kth_value', 'compute_metric'}
    pfns = [n for n in ast.parse(psrc).body if isinstance(n, ast.FunctionDef) and n.name in pwant]
    pns = {'np': np, 'trange': lambda n, desc='': range(n)}
    exec(compile(ast.Module(body=pfns, type_ignores=[]), 'precision_recall.py', 'exec'), pns)
    Manifold = namedtuple('Manifold', ['features', 'radii'])
    fr = (rng.randn(180, 24)).astype(np.float32)
    ff = (rng.randn(150, 24) * 0.9 + 0.35).astype(np.float32)
    for kk in (3, 5):
        mr = Manifold(fr, pns['distances2radii'](pns['compute_pairwise_distances'](fr), k=kk))
        mf = Manifold(ff, pns['distances2radii'](pns['compute_pairwise_distances'](ff), k=kk))
        out[f'pr/k{kk}'] = np.array([pns['compute_metric'](mr, ff), pns['compute_metric'](mf, fr)])
    out['pr/real'], out['pr/fake'] = fr, ff
    np.savez_compressed(os.path.join(OUT, 'fid.npz'), **out)
    print('fid.npz:', {k: float(v) for k, v in out.items() if k.endswith('/fid')})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    if a.only == 'fid':
        gen_fid()
        return
    torch.manual_seed(1)
    op, mpt = import_reference()
    todo = [a.only] if a.only else ['latents', 'ops', 'layers', 'small', 'full', 'ada']
    if 'latents' in todo:
        gen_latents()
    if 'ops' in todo:
        gen_ops(op)
    if 'layers' in todo:
        gen_layers(mpt)
    if 'small' in todo:
        gen_small(mpt)
    if 'full' in todo:
        gen_full(mpt)
    if 'ada' in todo:
        gen_ada()
    if not a.only:
        gen_fid()


if __name__ == '__main__':
    main()
