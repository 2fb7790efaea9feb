"""Diagnostic (GPU): image-gradient and activation-gradient errors vs the fp64 CPU oracle at 16 px, B=4."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_models import build
from oracle.model_ref import generator_ref, discriminator_ref
from oracle.train_ref import g_nonsaturating_loss_ref
from rick_amd.synth import synth_latents, synth_reals, synth_state_dict
from rick_amd.train import g_nonsaturating_loss
from tests.shapes import generator_shapes, discriminator_shapes
def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max())
for size, B in ((16, 4), (16, 2), (32, 4)):
    g, d = build(size)
    sg = {k: v.double() for k, v in synth_state_dict(generator_shapes(size)).items()}
    sd = {k: v.double() for k, v in synth_state_dict(discriminator_shapes(size)).items()}
    z = synth_latents(B, seed=size)
    fake_r, feats_r = generator_ref(sg, [z.double()], size=size, randomize_noise=False, return_feats=True)
    fake_r = fake_r.detach().requires_grad_(True)
    fp_r, feat_r = discriminator_ref(sd, fake_r, size=size)
    gl_r = g_nonsaturating_loss_ref(fp_r)
    gi_r = torch.autograd.grad(gl_r, [fake_r] + feat_r)
    fake = fake_r.detach().float().cuda().requires_grad_(True)
    fp, feat = d(fake)
    gl = g_nonsaturating_loss(fp)
    gi = torch.autograd.grad(gl, [fake] + feat)
    print(f'size {size} B {B}: loss rel {abs(float(gl)-float(gl_r))/abs(float(gl_r)):.2e}')
    print('   fwd feat errs', ['%.1e' % rel(a, b) for a, b in zip(feat, feat_r)])
    print('   grad wrt image %.2e ; wrt feats' % rel(gi[0], gi_r[0]), ['%.1e' % rel(a, b) for a, b in zip(gi[1:], gi_r[1:])])
