"""Diagnostic (GPU): per-op data-gradient errors at the 8x8 / 512-channel shapes, N = 2 vs 4."""
import sys, os, math
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd import op
from rick_amd.synth import synth_tensor
from oracle.ops_ref import upfirdn2d_ref, fused_leaky_relu_ref, make_blur_kernel
def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max())
k = make_blur_kernel([1, 3, 3, 1])
for N in (2, 4):
    C = 512
    for name, H, ks, s, p in (('c3s1', 8, 3, 1, 1), ('c3s2', 9, 3, 2, 0), ('c1s2', 7, 1, 2, 0)):
        x = synth_tensor(f'd/{name}/x{N}', (N, C, H, H)); w = synth_tensor(f'd/{name}/w', (C, C, ks, ks))
        xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
        yr = F.conv2d(xr, wr * 0.01, stride=s, padding=p)
        gy = synth_tensor(f'd/{name}/gy{N}', yr.shape)
        gxr, gwr = torch.autograd.grad(yr, (xr, wr), gy.double())
        xd, wd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
        y = op.conv2d(xd, wd, s, p, wscale=0.01)
        gx, gw = torch.autograd.grad(y, (xd, wd), gy.cuda())
        print(f'N={N} {name}: fwd {rel(y, yr):.1e} dgrad {rel(gx, gxr):.1e} wgrad {rel(gw, gwr):.1e}')
    for name, pad in (('blur22', (2, 2)), ('blur11', (1, 1))):
        x = synth_tensor(f'd/{name}/x{N}', (N, C, 8, 8))
        xr = x.double().requires_grad_(True)
        yr = upfirdn2d_ref(xr, k.double(), pad=pad)
        gy = synth_tensor(f'd/{name}/gy{N}', yr.shape)
        (gxr,) = torch.autograd.grad(yr, xr, gy.double())
        xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = op.upfirdn2d(xd, k.cuda(), pad=pad)
        (gx,) = torch.autograd.grad(y, xd, gy.cuda())
        print(f'N={N} {name}: fwd {rel(y, yr):.1e} grad {rel(gx, gxr):.1e}')
    x = synth_tensor(f'd/act/x{N}', (N, C, 8, 8)); b = synth_tensor('d/act/b', (C,))
    xr, br = x.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = fused_leaky_relu_ref(xr, br); gy = synth_tensor(f'd/act/gy{N}', yr.shape)
    gxr, gbr = torch.autograd.grad(yr, (xr, br), gy.double())
    xd, bd = x.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = op.fused_leaky_relu(xd, bd); gx, gb = torch.autograd.grad(y, (xd, bd), gy.cuda())
    print(f'N={N} act: fwd {rel(y, yr):.1e} gx {rel(gx, gxr):.1e} gb {rel(gb, gbr):.1e}')
    a = synth_tensor(f'd/as/a{N}', (N, C, 4, 4)); bb = synth_tensor(f'd/as/b{N}', (N, C, 4, 4))
    ad, bd2 = a.cuda().requires_grad_(True), bb.cuda().requires_grad_(True)
    y = op.add_scale(ad, bd2, 0.7071); g = synth_tensor(f'd/as/g{N}', y.shape)
    ga, gb2 = torch.autograd.grad(y, (ad, bd2), g.cuda())
    print(f'N={N} add_scale: fwd {rel(y, (a + bb) * 0.7071):.1e} ga {rel(ga, g * 0.7071):.1e} gb {rel(gb2, g * 0.7071):.1e}')
    x = synth_tensor(f'd/mb/x{N}', (N, C, 4, 4)); xr = x.double().requires_grad_(True)
    from oracle.ops_ref import minibatch_stddev_ref
    yr = minibatch_stddev_ref(xr); g = synth_tensor(f'd/mb/g{N}', yr.shape)
    (gr,) = torch.autograd.grad(yr, xr, g.double())
    xd = x.cuda().requires_grad_(True); y = op.minibatch_stddev(xd); (gd,) = torch.autograd.grad(y, xd, g.cuda())
    print(f'N={N} mbstd: fwd {rel(y, yr):.1e} grad {rel(gd, gr):.1e}')
