"""Per-key sum(g^2) error of the HIP path vs the reference's fp32 goldens at 256 px (and vs its fp64 run): the numbers
behind tests/test_gpu_models.py::check_grad2's bounds."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_models import build, grad2
from rick_amd.synth import synth_reals
from rick_amd.train import d_logistic_loss, g_nonsaturating_loss
G = lambda n: np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', n + '.npz'))
gold, g64 = G('full256'), G('spread256')
lat = torch.from_numpy(np.concatenate([G('noise_latents')[f'noise_{j:04d}'] for j in range(2)], 0)).cuda()
g, d = build(256)
real = synth_reals(2, size=256, seed=256).cuda()
fake, _ = g([lat], randomize_noise=False)
fp, _ = d(fake); rp, _ = d(real)
gd = torch.autograd.grad(d_logistic_loss(rp, fp), [p for _, p in d.named_parameters()], retain_graph=True, allow_unused=True)
gg = torch.autograd.grad(g_nonsaturating_loss(fp), [p for _, p in g.named_parameters()], allow_unused=True)
for tag, named, grads in (('d_grad2', list(d.named_parameters()), gd), ('g_grad2', list(g.named_parameters()), gg)):
    got = grad2(named, grads)
    for ref, nm in ((gold, 'f256'), (g64, 'f256_f64')):
        rels = sorted((abs(v - float(ref[f'{nm}/{tag}/{k}'])) / float(ref[f'{nm}/{tag}/{k}']), k) for k, v in got.items()
                      if float(ref[f'{nm}/{tag}/{k}']) > 0)
        print(tag, 'vs', nm, 'median %.2e' % rels[len(rels) // 2][0], 'p90 %.2e' % rels[int(len(rels) * 0.9)][0],
              'worst', ['%.1e %s' % r for r in rels[-4:]])
