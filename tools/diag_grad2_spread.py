"""256-px parity numbers of the HIP path against the reference's fp32 goldens (full256.npz) and its fp64 run
(spread256.npz), next to the reference's OWN fp32-vs-fp64 spread: first order (image, logits, losses, per-key grad^2)
and second order (R1, path length, their per-key grad^2).  The numbers behind tests/test_gpu_models.py's bounds."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd import op  # noqa: E402
from rick_amd.synth import synth_reals, synth_tensor  # noqa: E402
from rick_amd.train import d_logistic_loss, d_r1_loss, g_nonsaturating_loss, g_path_regularize  # noqa: E402
from tests.test_gpu_models import build, grad2  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = lambda n: np.load(os.path.join(ROOT, 'tests', 'golden', n + '.npz'))  # noqa: E731
g32, g64 = G('full256'), G('spread256')
lat = torch.from_numpy(np.concatenate([G('noise_latents')[f'noise_{j:04d}'] for j in range(2)], 0)).cuda()
g, d = build(256)
real = synth_reals(2, size=256, seed=256).cuda()
gp, dp = list(g.named_parameters()), list(d.named_parameters())


def scalar(name, v):
    v = np.asarray(v.detach().double().cpu())
    a, b = np.asarray(g32[f'f256/{name}'], dtype=np.float64), np.asarray(g64[f'f256_f64/{name}'], dtype=np.float64)
    den = np.abs(b).max()
    print(f'{name:14s} hip-vs-fp64 {np.abs(v - b).max() / den:.2e}   hip-vs-ref32 {np.abs(v - a).max() / den:.2e}   '
          f'ref32-vs-fp64 {np.abs(a - b).max() / den:.2e}')


def keys(tag, named, grads):
    got = grad2(named, grads)
    rows = []
    for k, v in got.items():
        r64 = float(g64[f'f256_f64/{tag}/{k}'])
        if r64 <= 0:
            continue
        r32 = float(g32[f'f256/{tag}/{k}'])
        rows.append((abs(v - r64) / r64, abs(r32 - r64) / r64, k))
    hip = sorted(r[0] for r in rows)
    ref = sorted(r[1] for r in rows)
    worst = sorted(rows)[-3:]
    print(f'{tag:10s} hip-vs-fp64 median {hip[len(hip) // 2]:.2e} p90 {hip[int(len(hip) * .9)]:.2e} max {hip[-1]:.2e} | '
          f'ref32-vs-fp64 median {ref[len(ref) // 2]:.2e} p90 {ref[int(len(ref) * .9)]:.2e} max {ref[-1]:.2e} | worst hip keys '
          + ', '.join(f'{k} {a:.1e} (ref {b:.1e})' for a, b, k in worst))


fake, _ = g([lat], randomize_noise=False)
idx = torch.from_numpy(g32['f256/img_idx']).cuda()
scalar('img_samples', fake.reshape(2, -1)[:, idx])
fp, _ = d(fake)
rp, _ = d(real)
scalar('fake_pred', fp)
scalar('real_pred', rp)
d_loss, g_loss = d_logistic_loss(rp, fp), g_nonsaturating_loss(fp)
scalar('d_loss', d_loss)
scalar('g_loss', g_loss)
gd = torch.autograd.grad(d_loss, [p for _, p in dp], retain_graph=True, allow_unused=True)
gg = torch.autograd.grad(g_loss, [p for _, p in gp], allow_unused=True)
keys('d_grad2', dp, gd)
keys('g_grad2', gp, gg)
with op.second_order():
    real_r = real.clone().requires_grad_(True)
    rpr, _ = d(real_r)
    r1 = d_r1_loss(rpr, real_r)
    scalar('r1', r1)
    gr1 = torch.autograd.grad(10 / 2 * r1 * 16 + 0 * rpr[0].sum(), [p for _, p in dp], allow_unused=True)
    keys('r1_grad2', dp, gr1)
    img, lt = g([lat[:1]], return_latents=True, randomize_noise=False)
    pl_noise = synth_tensor('plnoise/256', img.shape).cuda()
    pen, _, lens = g_path_regularize(img, lt, 0, noise=pl_noise)
    scalar('pl_lengths', lens)
    scalar('pl_loss', pen)
    gpl = torch.autograd.grad(8 * pen + 0 * img[0, 0, 0, 0], [p for _, p in gp], allow_unused=True)
    keys('pl_grad2', gp, gpl)
