"""Error of the fp16x3 MFMA convolution family vs fp64 (GPU box): magnitudes far from 1, a dynamic-range case whose
lo parts are fp16 subnormals (tells whether v_mfma_f32_16x16x32_f16 flushes them), heavy tails."""
import math
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, '.')
from rick_amd import op  # noqa: E402

DEV = 'cuda:0'
torch.manual_seed(0)


def rel(a, b):
    return float((a.double().cpu() - b).abs().max() / b.abs().max())


def run(tag, x, w, s=1, p=1, gy_scale=1.0):
    wscale = 1 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3])
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr * wscale, stride=s, padding=p)
    gy = torch.randn(yr.shape) * gy_scale
    gxr, gwr = torch.autograd.grad(yr, (xr, wr), gy.double())
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = op.conv2d(xd, wd, s, p, wscale=wscale)
    gx, gw = torch.autograd.grad(y, (xd, wd), gy.to(DEV))
    y32 = F.conv2d(x, w * wscale, stride=s, padding=p)
    print(f'{tag:34s} fprop {rel(y, yr.detach()):.2e} (cpu fp32 {rel(y32, yr.detach()):.2e})  dgrad {rel(gx, gxr):.2e}  wgrad {rel(gw, gwr):.2e}',
          flush=True)


for sc in (1.0, 1e-7, 3e4, 1e-20):
    run(f'64->128 16x16 x*{sc:g}', torch.randn(2, 64, 16, 16) * sc, torch.randn(128, 64, 3, 3), gy_scale=sc)
run('512->512 16x16', torch.randn(2, 512, 16, 16), torch.randn(512, 512, 3, 3))
run('128->128 64x64', torch.randn(1, 128, 64, 64), torch.randn(128, 128, 3, 3))
run('s2 64->128 33x33', torch.randn(1, 64, 33, 33), torch.randn(128, 64, 3, 3), s=2, p=0)
run('w*1e-6', torch.randn(2, 64, 16, 16), torch.randn(128, 64, 3, 3) * 1e-6)
# dynamic range: channel 0 carries the maximum (its weights are zero), everything else is 2^-10 of it -> scaled values
# ~ 0.016, hi normal, lo ~ 8e-6 = fp16 subnormal.  ~2e-6 if the MFMA keeps subnormal inputs, ~2e-4 if it flushes them.
x = torch.randn(2, 64, 16, 16) * 2.0 ** -10
x[:, 0] = torch.randn(2, 16, 16)
w = torch.randn(128, 64, 3, 3)
w[:, 0] = 0
run('subnormal-lo probe', x, w)
# heavy tails: cubed normals (max / median ~ 300)
run('heavy tails x^3', torch.randn(2, 64, 16, 16) ** 3, torch.randn(128, 64, 3, 3) ** 3)
# sparse outliers: one element 1e4 x the rest (the sample may or may not see it)
x = torch.randn(2, 64, 16, 16)
x[1, 37, 5, 9] = 3e3
run('one outlier 3e3', x, torch.randn(128, 64, 3, 3))
x[1, 37, 5, 9] = 1e5
run('one outlier 1e5 (overflow probe)', x, torch.randn(128, 64, 3, 3))
# transposed stride 2 (convt2 kernel)
xr = torch.randn(2, 64, 16, 16)
w = torch.randn(32, 64, 3, 3)
ref = F.conv_transpose2d(xr.double(), w.double().transpose(0, 1) * 0.1, stride=2)
y = op.conv_transpose2d(xr.to(DEV), w.to(DEV), 2, 0, wscale=0.1)
print(f'convT2 64->32 16x16                fprop {rel(y, ref):.2e}')
ref = F.conv_transpose2d(xr.double() * 1e-9, w.double().transpose(0, 1) * 0.1, stride=2)
y = op.conv_transpose2d((xr * 1e-9).to(DEV), w.to(DEV), 2, 0, wscale=0.1)
print(f'convT2 64->32 16x16 x*1e-9         fprop {rel(y, ref):.2e}')
