"""Device time of the 1x1 weight gradients of the discriminator's skip convolutions (both operands split images, batch B)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.op import conv as cv, split as sp
from tools.bench_elem import timeit
B = int(os.environ.get('B', 8))
for ci, co, r in [(128, 256, 128), (256, 512, 64), (512, 512, 32), (512, 512, 16)]:
    x = torch.randn(B, ci, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
    xs, gs = sp.split_pack(x), sp.split_pack(gy)
    ref = torch.einsum('noyx,niyx->oi', gy.double(), x.double()).float()
    t0 = timeit(lambda: cv._wgrad_launch(gy, x, 1, 1, 1, 0))
    t1 = timeit(lambda: cv._wgrad_launch(None, None, 1, 1, 1, 0, a_split=gs, b_split=xs))
    e0 = float((cv._wgrad_launch(gy, x, 1, 1, 1, 0).view(co, ci) - ref).abs().max() / ref.abs().max())
    e1 = float((cv._wgrad_launch(None, None, 1, 1, 1, 0, a_split=gs, b_split=xs).view(co, ci) - ref).abs().max() / ref.abs().max())
    print(f'wgrad 1x1 {ci:4d}x{co:4d} @{r:3d} N{B}: fp32 {t0*1e6:7.1f} us (err {e0:.1e}) | both split {t1*1e6:7.1f} us (err {e1:.1e})')
