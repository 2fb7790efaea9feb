"""Diagnostic (GPU): per-key relative error of parameter-gradient norms vs the goldens."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_models import build, grad2
from rick_amd.synth import synth_latents, synth_reals
from rick_amd.train import d_logistic_loss, g_nonsaturating_loss
gold = np.load('tests/golden/small.npz')
for tag, size, B in (('s32_f64', 32, 2), ('s16_f64', 16, 4)):
    g, d = build(size)
    z = synth_latents(B, seed=size).cuda(); real = synth_reals(B, size=size, seed=size).cuda()
    gp, dp = list(g.named_parameters()), list(d.named_parameters())
    fake, _ = g([z], randomize_noise=False)
    fp, _ = d(fake); rp, _ = d(real)
    dl = d_logistic_loss(rp, fp); gl = g_nonsaturating_loss(fp)
    gd = torch.autograd.grad(dl, [p for _, p in dp], retain_graph=True)
    gg = torch.autograd.grad(gl, [p for _, p in gp], retain_graph=True, allow_unused=True)
    for pre, named, grads in (('d_grad2', dp, gd), ('g_grad2', gp, gg)):
        got = grad2(named, grads)
        rows = []
        for k, v in got.items():
            ref = float(gold[f'{tag}/{pre}/{k}'])
            rows.append((abs(v - ref) / (ref + 1e-300), k, v, ref))
        rows.sort(reverse=True)
        print(tag, pre, 'worst:')
        for r in rows[:8]:
            print('   %.2e  %-40s got %.4e ref %.4e' % r)
        print('   median rel %.2e' % np.median([r[0] for r in rows]))
