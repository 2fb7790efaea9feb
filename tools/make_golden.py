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

usage: python tools/make_golden.py [--only ops|layers|small|full|latents|fid|ada|rick|spread|eval]
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


def model_case(mpt, size, B, dtype, out, tag, latents=None, full_arrays=False, store=None):
    np32 = store or globals()['np32']     # (gen_spread keeps the fp64 run's values in fp64)
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


# ------------------------------------------------------------- RICK loop (rows Q, K, O, H)
TRAIN_PY = os.path.join(REF, 'train_dynamic_update_prune.py')


def ref_functions():
    """The reference's own top-level functions of train_dynamic_update_prune.py (:63-144), cut out of the file with
    `ast` (the script itself needs lmdb / torchvision / CUDA at import) and executed unchanged."""
    import ast
    want = {'requires_grad', 'accumulate', 'd_logistic_loss', 'd_r1_loss', 'g_nonsaturating_loss', 'g_path_regularize',
            'zero_idx_merge'}
    src = open(TRAIN_PY).read()
    fns = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert {f.name for f in fns} == want
    ns = {'torch': torch, 'F': torch.nn.functional, 'autograd': torch.autograd, 'math': math, 'np': np}
    exec(compile(ast.Module(body=fns, type_ignores=[]), TRAIN_PY, 'exec'), ns)
    return ns


def ref_slice(first, last, head, tail):
    """Lines first..last (1-based, inclusive) of the reference training script, dedented, compiled with the file's own
    line numbers.  `head` / `tail` are substrings the first / last line must contain (guards against a shifted file)."""
    import textwrap
    lines = open(TRAIN_PY).read().split('\n')
    assert head in lines[first - 1] and tail in lines[last - 1], (lines[first - 1], lines[last - 1])
    body = textwrap.dedent('\n'.join(lines[first - 1:last]))
    return compile('\n' * (first - 1) + body, TRAIN_PY, 'exec')


def key_samples(key, n, count=64):
    """Fixed pseudo-random element indices of a tensor with n elements, seeded by its key (tests rebuild them)."""
    import zlib
    return np.random.RandomState(zlib.crc32(key.encode()) & 0x7fffffff).randint(0, n, size=min(count, n))


def summarise(out, prefix, named):
    for k, t in named:
        a = t.detach().double().reshape(-1)
        out[f'{prefix}/{k}/sum'] = np.float64(a.sum())
        out[f'{prefix}/{k}/sq'] = np.float64((a * a).sum())
        out[f'{prefix}/{k}/samples'] = a[torch.from_numpy(key_samples(k, a.numel()))].numpy().astype(np.float32)


def fisher_sweep_ref(mpt, fns, g_ema, d_ema, latents, reals):
    """The sample loop of the Fisher sweep (:225-263) with the reference's estimate_fisher and loss functions; the
    accumulation lines are the reference's (numpy, on the host)."""
    filter_fisher_g, filter_fisher_d = dict(), dict()
    for j, (noise_fisher, real_img_fisher) in enumerate(zip(latents, reals)):
        for fisher_idx in range(noise_fisher.size()[0]):
            g_ema.zero_grad()
            d_ema.zero_grad()
            fake_img_fisher, _ = g_ema([(noise_fisher.data)[fisher_idx].view(1, -1)], randomize_noise=False)
            batch_1_real_img = (real_img_fisher.data)[fisher_idx].view(1, 3, 256, 256)
            fake_pred_fisher, _ = d_ema(fake_img_fisher)
            real_pred_fisher, _ = d_ema(batch_1_real_img)
            g_loss_fisher = fns['g_nonsaturating_loss'](fake_pred_fisher)
            d_loss_fisher = fns['d_logistic_loss'](real_pred_fisher, fake_pred_fisher)
            _, est_fisher_info_g = g_ema.estimate_fisher(loglikelihood=g_loss_fisher)
            _, est_fisher_info_d = d_ema.estimate_fisher(loglikelihood=d_loss_fisher)
            for key in est_fisher_info_g:
                if j == 0 and fisher_idx == 0:
                    filter_fisher_g[key] = est_fisher_info_g[key].detach().cpu().numpy()
                else:
                    filter_fisher_g[key] += est_fisher_info_g[key].detach().cpu().numpy()
            for key in est_fisher_info_d:
                if j == 0 and fisher_idx == 0:
                    filter_fisher_d[key] = est_fisher_info_d[key].detach().cpu().numpy()
                else:
                    filter_fisher_d[key] += est_fisher_info_d[key].detach().cpu().numpy()
        print('  fisher sample', j, flush=True)
    return filter_fisher_g, filter_fisher_d


def gen_rick(mpt):
    """Rows Q / K / O / H of SURVEY §8: the reference's OWN decision block (277-393, executed as a source slice on
    Fisher dictionaries produced by its estimate_fisher), and its OWN loop body (:401-589, executed slice by slice:
    D step, R1, G step, path length, with torch.optim.Adam built by the slice :887-931) at 256 px, batch 4 — the
    benchmarked configuration.  Deviations from the script, all forced by the container: CPU instead of .cuda(),
    `randomize_noise=False` (noise buffers instead of fresh normal_ draws: the device RNG cannot be matched), fixed
    latents instead of mixing_noise's RNG, torch.randn_like of the path-length noise replaced by a seeded tensor."""
    fns = ref_functions()
    out = {}
    size, B = 256, 4
    dtype = torch.float32
    torch.set_num_threads(os.cpu_count() or 8)
    g, d = build(mpt, size, dtype)
    g_ema, d_ema = build(mpt, size, dtype)
    g_ema.eval()
    d_ema.eval()

    # ---------------- rows F / Q: two Fisher sweeps of 2 samples each on the shipped _noise latents
    lat = [torch.load(os.path.join(REF, '_noise', f'{j:04d}.pt')) for j in range(4)]
    reals = [synth_reals(1, size=size, seed=700 + j) for j in range(4)]
    from types import SimpleNamespace
    decide = ref_slice(277, 393, '# Obtain the quantile values', 'zero_filter_idx_d = zero_idx_merge')
    sweeps = []
    for sw in range(2):
        fg, fd = fisher_sweep_ref(mpt, fns, g_ema, d_ema, lat[2 * sw:2 * sw + 2], reals[2 * sw:2 * sw + 2])
        for key in fg:                     # :266-269 with num_fisher_img = 2, batch = 4
            fg[key] /= (2 * B)
        for key in fd:
            fd[key] /= (2 * B)
        sweeps.append((fg, fd))
        # per-filter FIM vectors in the reference's own expressions (:281-297, :339-351): the decision functions of
        # the build are fed these and must reproduce the index sets below exactly
        for k in range(12):
            out[f'fim{sw}/g_conv/convs.{k}.conv.weight'] = fg[f'convs.{k}.conv.weight'].mean(axis=(0, 2, 3, 4))
            out[f'fim{sw}/g_fc/convs.{k}.conv.modulation.weight'] = (
                fg[f'convs.{k}.conv.modulation.weight'].mean(axis=1) + fg[f'convs.{k}.conv.modulation.bias']) / 2
        for b in range(1, 7):
            for li in range(2):
                wk, bk = f'convs.{b}.conv{li + 1}.{li}.weight', f'convs.{b}.conv{li + 1}.{li + 1}.bias'
                out[f'fim{sw}/d/{wk}'] = (fd[wk].mean(axis=(1, 2, 3)) + fd[bk]) / 2
            sk = f'convs.{b}.skip.1.weight'
            out[f'fim{sw}/d/{sk}'] = fd[sk].mean(axis=(1, 2, 3))
    masks = None
    for qtag, fq, pq in (('q40', 40.0, 0.1), ('q85', 85.0, 0.075)):     # README recipes (Babies / AFHQ-Cat)
        args = SimpleNamespace(fisher_quantile=fq, prune_quantile=pq, warmup_iter=250)
        ns = {'np': np, 'args': args, 'zero_idx_merge': fns['zero_idx_merge']}
        for sw, (fg, fd) in enumerate(sweeps):
            ns.update(filter_fisher_g=fg, filter_fisher_d=fd, i=250 + 50 * sw)
            exec(decide, ns)
            for nm in ('idx_freeze_g', 'idx_ft_g', 'idx_prune_g', 'idx_freeze_d', 'idx_ft_d', 'idx_prune_d',
                       'zero_filter_idx_g', 'zero_filter_idx_d'):
                for key, idx in ns[nm].items():
                    out[f'{qtag}/s{sw}/{nm}/{key}'] = np.asarray(idx, dtype=np.int32)
            for nm in ('cutline_g_conv', 'pruneline_g_conv', 'cutline_g_fc', 'pruneline_g_fc', 'cutline_d_conv',
                       'pruneline_d_conv'):
                out[f'{qtag}/s{sw}/{nm}'] = np.float64(ns[nm])
            if qtag == 'q40' and sw == 0:
                masks = {nm: dict(ns[nm]) for nm in ('idx_freeze_g', 'idx_freeze_d', 'zero_filter_idx_g', 'zero_filter_idx_d')}
    del sweeps

    # ---------------- rows L / R1 / PL / K / O / H: the loop body, slice by slice
    class Net:
        """What the script's `generator` / `discriminator` names are bound to: the module (called with fixed noise
        buffers), reachable through `.module` like under nn.DataParallel (:941-944)."""

        def __init__(self, m, **kw):
            self.__dict__['m'], self.__dict__['kw'] = m, kw

        def __call__(self, *a, **k):
            return self.m(*a, **{**self.kw, **k})

        def __getattr__(self, name):
            return self.m if name == 'module' else getattr(self.m, name)

    class TorchProxy:
        """`torch` as the path-length function sees it: randn_like returns the seeded tensor the tests share."""

        def __init__(self, noise):
            self.noise = noise

        def randn_like(self, t):
            assert t.shape == self.noise.shape
            return self.noise.to(t.dtype)

        def __getattr__(self, name):
            return getattr(torch, name)

    sys.path.insert(0, REF)
    import distributed as ref_dist                       # reference helpers (no process group: early-return branches)
    args = SimpleNamespace(size=size, batch=B, latent=512, mixing=0.9, lr=0.002, r1=10, path_regularize=2,
                           path_batch_shrink=2, d_reg_every=16, g_reg_every=4, warmup_iter=0, augment=False,
                           augment_p=0)
    generator, discriminator = Net(g, randomize_noise=False), Net(d)
    z = {'d': synth_latents(B, seed=901), 'g': synth_latents(B, seed=902), 'plr': synth_latents(B // 2, seed=903)}
    real_img = synth_reals(B, size=size, seed=904)
    pl_noise = synth_tensor('plnoise/rick256', (B // 2, 3, size, size))
    fns['torch'] = TorchProxy(pl_noise)                  # g_path_regularize's global `torch`
    order = iter(['g', 'plr'])
    ns = dict(fns)
    ns.update(args=args, generator=generator, discriminator=discriminator, optim=torch.optim, device='cpu', i=16,
              loss_dict={}, mean_path_length=0, noise=[z['d']], real_img=real_img,
              mixing_noise=lambda batch, latent, prob, device: [z[next(order)][:batch]],
              reduce_sum=ref_dist.reduce_sum, get_world_size=ref_dist.get_world_size, torch=torch,
              idx_freeze_g=masks['idx_freeze_g'], idx_freeze_d=masks['idx_freeze_d'],
              zero_filter_idx_g=masks['zero_filter_idx_g'], zero_filter_idx_d=masks['zero_filter_idx_d'])
    exec(ref_slice(887, 931, 'g_reg_ratio = args.g_reg_every', ')'), ns)          # Adam over the probe parameter lists
    gp, dp = list(g.named_parameters()), list(d.named_parameters())
    g_opt_keys = [k for k, _ in gp if 'convs' in k]
    d_opt_keys = [k for k, _ in dp if ('convs' in k and 'convs.0' not in k) or 'final' in k]
    assert len(ns['g_probe_params']) == len(g_opt_keys) and len(ns['d_probe_params']) == len(d_opt_keys)

    def grads(named, keys):
        return [(k, p.grad) for k, p in named if k in keys and p.grad is not None]

    print('  D step', flush=True)
    exec(ref_slice(401, 438, 'fake_img, _ = generator(noise)', 'torch.cuda.empty_cache()'), ns)
    out['step/d_loss'] = np32(ns['d_loss'])
    out['step/real_pred'], out['step/fake_pred'] = np32(ns['real_pred']), np32(ns['fake_pred'])
    summarise(out, 'step/d_grad', grads(dp, d_opt_keys))
    summarise(out, 'step/d_param', [(k, p) for k, p in dp if k in d_opt_keys])
    print('  R1 step', flush=True)
    exec(ref_slice(461, 495, '# using r1_loss', 'loss_dict["r1"] = r1_loss'), ns)
    out['step/r1_loss'] = np32(ns['r1_loss'])
    summarise(out, 'step/r1_grad', grads(dp, d_opt_keys))
    summarise(out, 'step/r1_param', [(k, p) for k, p in dp if k in d_opt_keys])
    ns['real_img'] = real_img            # the slice set requires_grad on it; the G step does not use it
    print('  G step', flush=True)
    exec(ref_slice(497, 540, '# adversarial training G', 'torch.cuda.empty_cache()'), ns)
    out['step/g_loss'] = np32(ns['loss_dict']['g'])
    summarise(out, 'step/g_grad', grads(gp, g_opt_keys))
    summarise(out, 'step/g_param', [(k, p) for k, p in gp if k in g_opt_keys])
    print('  path-length step', flush=True)
    ns.update(g_loss=None, d_loss=None, fake_img=None, fake_pred=None, real_pred=None)
    exec(ref_slice(546, 589, 'g_regularize = i % args.g_reg_every == 0', ')'), ns)
    out['step/path_loss'] = np32(ns['path_loss'])
    out['step/path_lengths'] = np32(ns['path_lengths'])
    out['step/mean_path_length'] = np32(ns['mean_path_length'])
    summarise(out, 'step/pl_grad', grads(gp, g_opt_keys))
    summarise(out, 'step/pl_param', [(k, p) for k, p in gp if k in g_opt_keys])
    # EMA (:180, :697-698)
    accum = 0.5 ** (32 / (10 * 1000))
    fns['accumulate'](g_ema, g, accum)
    fns['accumulate'](d_ema, d, accum)
    summarise(out, 'step/g_ema', list(g_ema.named_parameters()))
    summarise(out, 'step/d_ema', list(d_ema.named_parameters()))
    np.savez_compressed(os.path.join(OUT, 'rick256.npz'), **out)
    print('rick256.npz', len(out), 'arrays')


def gen_spread(mpt):
    """The 256-px case of gen_full once more in fp64 — every quantity of model_case incl. the second-order ones (R1,
    path length and their parameter gradients): how far the REFERENCE's own fp32 run sits from its fp64 run, per
    parameter key (LeakyReLU sign flips of ~0 pre-activations) — the yardstick for the GPU tests' bounds."""
    out = {}
    lat = torch.cat([torch.load(os.path.join(REF, '_noise', f'{j:04d}.pt')) for j in range(2)], 0)
    model_case(mpt, 256, 2, torch.float64, out, 'f256_f64', latents=lat,
               store=lambda t: t.detach().to(torch.float64).numpy().copy())
    del out['f256_f64/img_idx']          # (same indices as full256.npz)
    np.savez_compressed(os.path.join(OUT, 'spread256.npz'), **out)
    print('spread256.npz', len(out), 'arrays')


def gen_eval(mpt):
    """BASELINE config 4 at its own batch size: the evaluator's sampling loop (gan_training/eval.py:34-41) draws
    n_sample_store = 25 latents per g_ema call.  Two such calls on the first 50 rows of the shipped noise.pt with the
    registered noise buffers (randomize_noise=False: the device RNG cannot reproduce the reference's draws), fp32 like the
    reference: per-image channel means / stds of all 50 images and 4096 sampled pixels of six of them."""
    out = {}
    g, _ = build(mpt, 256, torch.float32)
    z = torch.load(os.path.join(REF, 'noise.pt'))[:50]
    idx = torch.from_numpy(np.random.RandomState(0).randint(0, 3 * 256 * 256, size=4096))
    means, stds, picks = [], [], {}
    with torch.no_grad():
        for b in range(2):
            img, _ = g([z[25 * b:25 * (b + 1)]], randomize_noise=False)
            means.append(img.mean(dim=(2, 3)))
            stds.append(img.std(dim=(2, 3)))
            for i in (0, 12, 24):
                picks[25 * b + i] = img[i].reshape(-1)[idx]
            print('  eval batch', b, flush=True)
    out['e256/img_mean'] = np32(torch.cat(means))
    out['e256/img_std'] = np32(torch.cat(stds))
    out['e256/img_idx'] = idx.numpy()
    out['e256/picks'] = np.array(sorted(picks), dtype=np.int64)
    out['e256/img_samples'] = np32(torch.stack([picks[k] for k in sorted(picks)]))
    np.savez_compressed(os.path.join(OUT, 'eval256.npz'), **out)
    print('eval256.npz', len(out), 'arrays')


def check_shapes(mpt):
    """tests/shapes.py (the state_dict contract the build's modules are asserted against) == the reference modules'
    state_dict() keys and shapes, at the three sizes the tests use."""
    from tests.shapes import discriminator_shapes, generator_shapes
    for size in (16, 32, 256):
        g = mpt.Generator(size, 512, 8, channel_multiplier=2)
        d = mpt.Discriminator(size, channel_multiplier=2)
        assert {k: tuple(v.shape) for k, v in g.state_dict().items()} == generator_shapes(size), size
        assert {k: tuple(v.shape) for k, v in d.state_dict().items()} == discriminator_shapes(size), size
    print('tests/shapes.py == reference state_dict() at 16 / 32 / 256 px')


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
    todo = [a.only] if a.only else ['latents', 'ops', 'layers', 'small', 'full', 'ada', 'rick', 'spread', 'eval']
    check_shapes(mpt)
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
    if 'rick' in todo:
        gen_rick(mpt)
    if 'spread' in todo:
        gen_spread(mpt)
    if 'eval' in todo:
        gen_eval(mpt)
    if not a.only:
        gen_fid()


if __name__ == '__main__':
    main()
