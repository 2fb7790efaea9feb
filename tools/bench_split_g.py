"""Generator-shaped launches (batch 4, per-(image, channel) style scales): fp32 operands with the scales applied while staging
vs split images with the scales folded in by the producer."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.op import conv as cv, split as sp
from tools.bench_conv_util import timeit
B = int(os.environ.get('B', 4))
for ci, co, r in [(512, 512, 64), (256, 256, 128), (128, 128, 256), (512, 512, 32)]:
    x = torch.randn(B, ci, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 3, 3, device='cuda')
    si, so = torch.rand(B, ci, device='cuda') + 0.5, torch.rand(B, co, device='cuda') + 0.5
    wp, wpT = cv._pack(w, 1.0), cv._pack(w.transpose(0, 1), 1.0)
    xs, gs = sp.split_pack(x * si.view(B, ci, 1, 1)), sp.split_pack(gy * so.view(B, co, 1, 1))
    flops = 2.0 * B * r * r * ci * co * 9
    t0 = timeit(lambda: cv._conv_launch(x, wp, co, 3, 3, 1, 1, iscale=si, oscale=so))
    t1 = timeit(lambda: cv._conv_launch(None, wp, co, 3, 3, 1, 1, oscale=so, x_split=xs))
    t2 = timeit(lambda: cv._convT_launch(gy, wpT, ci, 3, 3, 1, 1, (r, r), iscale=so))
    t3 = timeit(lambda: cv._convT_launch(None, wpT, ci, 3, 3, 1, 1, (r, r), x_split=gs))
    t4 = timeit(lambda: cv._wgrad_launch(gy, x, 3, 3, 1, 1, ascale=so, bscale=si))
    t5 = timeit(lambda: cv._wgrad_launch(None, None, 3, 3, 1, 1, a_split=gs, b_split=xs))
    print(f's1 {ci:4d}->{co:4d} @{r:3d} N{B}: fprop {t0*1e6:6.1f} -> {t1*1e6:6.1f} us ({t0/t1:.2f}x) | dgrad {t2*1e6:6.1f} -> {t3*1e6:6.1f} ({t2/t3:.2f}x) | wgrad {t4*1e6:6.1f} -> {t5*1e6:6.1f} ({t4/t5:.2f}x)')
for ci, co, r in [(512, 512, 32), (512, 256, 64), (256, 128, 128)]:     # upsampling layers: convT2 fprop, s2 conv dgrad, s2 wgrad
    x = torch.randn(B, ci, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
    ro = 2 * r + 1
    gy = torch.randn(B, co, ro, ro, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 3, 3, device='cuda')
    si, so = torch.rand(B, ci, device='cuda') + 0.5, torch.rand(B, co, device='cuda') + 0.5
    wp, wpT = cv._pack(w, 1.0), cv._pack(w.transpose(0, 1), 1.0)
    xs, gs = sp.split_pack(x * si.view(B, ci, 1, 1)), sp.split_pack(gy * so.view(B, co, 1, 1))
    flops = 2.0 * B * r * r * ci * co * 9
    t0 = timeit(lambda: cv._convT_launch(x, wp, co, 3, 3, 2, 0, (ro, ro), iscale=si, oscale=so))
    t1 = timeit(lambda: cv._convT_launch(None, wp, co, 3, 3, 2, 0, (ro, ro), oscale=so, x_split=xs))
    t2 = timeit(lambda: cv._conv_launch(gy, wpT, ci, 3, 3, 2, 0, iscale=so))
    t3 = timeit(lambda: cv._conv_launch(None, wpT, ci, 3, 3, 2, 0, x_split=gs))
    t4 = timeit(lambda: cv._wgrad_launch(x, gy, 3, 3, 2, 0, ascale=si, bscale=so))
    t5 = timeit(lambda: cv._wgrad_launch(None, None, 3, 3, 2, 0, a_split=xs, b_split=gs))
    print(f'up {ci:4d}->{co:4d} @{r:3d} N{B}: fprop(ct2) {t0*1e6:6.1f} -> {t1*1e6:6.1f} us ({t0/t1:.2f}x) | dgrad(s2) {t2*1e6:6.1f} -> {t3*1e6:6.1f} ({t2/t3:.2f}x) | wgrad {t4*1e6:6.1f} -> {t5*1e6:6.1f} ({t4/t5:.2f}x)')
