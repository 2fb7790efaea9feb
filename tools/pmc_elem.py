"""A few launches of the HBM-bound kernels at one layer shape — target for `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE`."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.op import fused_act, misc
from rick_amd.op.upfirdn2d import upfirdn2d
B, c, r = 4, int(sys.argv[1]), int(sys.argv[2])
x = torch.randn(B, c, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
g = torch.randn_like(x)
bias = torch.randn(c, device='cuda')
noise = torch.randn(B, 1, r, r, device='cuda')
nw = torch.full((1,), 0.1, device='cuda')
k4 = torch.tensor([1., 3., 3., 1.], device='cuda')
k4 = torch.outer(k4, k4) / 64
xb = torch.randn(B, c, r + 1, r + 1, device='cuda').contiguous(memory_format=torch.channels_last)
for _ in range(4):
    y = fused_act.fused_noise_bias_act(x, bias, noise, nw)
    fused_act._ActAdjoint.apply(g, y, noise, 0.2, 2 ** 0.5, True, True)
    misc._hw_dot_raw(x, g)
    upfirdn2d(xb, k4 * 4, pad=(1, 1))
torch.cuda.synchronize()
