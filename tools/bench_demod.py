"""Latency of the demodulation path: fused kernels vs tensor algebra (GPU; rocprof-free event timing of a queued loop)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.op.modconv import demod_coeff, demod_coeff_fused
B, O, I = 4, 512, 512
w = torch.randn(O, I, 3, 3, device='cuda', requires_grad=True)
s = torch.randn(B, I, device='cuda', requires_grad=True)
gd = torch.randn(B, O, device='cuda')
def run(fn, n=200):
    for _ in range(20):
        d = fn(w, s, 0.01); torch.autograd.grad(d, [w, s], gd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        d = fn(w, s, 0.01); torch.autograd.grad(d, [w, s], gd)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print('composed fwd+bwd %.1f us   fused fwd+bwd %.1f us' % (run(demod_coeff), run(demod_coeff_fused)))
