"""Split-image operands vs the on-the-fly fp32 -> fp16 hi/lo split: accuracy against an fp64 reference and launch time of the
weight-gradient / forward kernels on the layer shapes of the 256-px discriminator (GPU)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.op import conv as cv, split as sp
from tools.bench_conv_util import timeit

B = int(os.environ.get('B', 8))
which = sys.argv[1:] or ['wgrad']
torch.manual_seed(0)
def ref_wgrad(gy, x, s, p):
    g64, x64 = gy.double().cpu(), x.double().cpu()
    w = torch.zeros(gy.shape[1], x.shape[1], 3, 3, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(x64, w, stride=s, padding=p)
    return torch.autograd.grad(y, w, g64)[0]

# accuracy on a small case (values spread over several octaves)
for s, p, r in [(1, 1, 32), (2, 0, 33)]:
    x = (torch.randn(2, 64, r, r, device='cuda') * torch.exp2(torch.randint(-6, 3, (2, 64, 1, 1), device='cuda').float())).contiguous(memory_format=torch.channels_last)
    ro = (r + 2 * p - 3) // s + 1
    gy = (torch.randn(2, 128, ro, ro, device='cuda') * 1e-4).contiguous(memory_format=torch.channels_last)
    ref = ref_wgrad(gy, x, s, p)
    xs, gs = sp.split_pack(x), sp.split_pack(gy)
    assert (sp.split_unpack(xs) - x).abs().max() <= 2.0 ** -21 * x.abs().max()
    for name, kw in [('fp32 operands', {}), ('gy split', dict(a_split=gs)), ('x split', dict(b_split=xs)), ('both split', dict(a_split=gs, b_split=xs))]:
        gw = cv._wgrad_launch(gy, x, 3, 3, s, p, **kw)
        err = float((gw.double().cpu() - ref).abs().max() / ref.abs().max())
        print(f'wgrad s{s} {name:14s} max err / max = {err:.2e}')

if 'wgrad' in which:
    for ci, co, r, s in [(512, 512, 64, 1), (256, 256, 128, 1), (128, 128, 256, 1), (512, 512, 32, 1), (128, 256, 256, 2), (256, 512, 128, 2), (512, 512, 64, 2)]:
        p = 1 if s == 1 else 0
        ri = r if s == 1 else r + 1
        ro = r if s == 1 else r // 2
        x = torch.randn(B, ci, ri, ri, device='cuda').contiguous(memory_format=torch.channels_last)
        gy = torch.randn(B, co, ro, ro, device='cuda').contiguous(memory_format=torch.channels_last)
        flops = 2.0 * B * ro * ro * ci * co * 9
        xs, gs = sp.split_pack(x), sp.split_pack(gy)
        t0 = timeit(lambda: cv._wgrad_launch(gy, x, 3, 3, s, p))
        t1 = timeit(lambda: cv._wgrad_launch(gy, x, 3, 3, s, p, a_split=gs))
        t2 = timeit(lambda: cv._wgrad_launch(gy, x, 3, 3, s, p, a_split=gs, b_split=xs))
        tp = timeit(lambda: sp.split_pack(x, xs.bound[0]))
        print(f'wgrad s{s} {ci:4d}x{co:4d} @{r:3d} N{B}: fp32 {t0*1e6:7.1f} us {flops/t0/1e12:6.1f} TF | gy split {t1*1e6:7.1f} us {flops/t1/1e12:6.1f} TF | '
              f'both {t2*1e6:7.1f} us {flops/t2/1e12:6.1f} TF ({t0/t2:.2f}x) | pack(x) pass {tp*1e6:6.1f} us')

def check(name, y, ref):
    err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    print(f'{name:32s} max err / max = {err:.2e}')

if 'conv' in which:
    F = torch.nn.functional
    x = (torch.randn(2, 64, 32, 32, device='cuda') * torch.exp2(torch.randint(-6, 3, (2, 64, 1, 1), device='cuda').float())).contiguous(memory_format=torch.channels_last)
    w = torch.randn(128, 64, 3, 3, device='cuda')
    xs, wp, wpT = sp.split_pack(x), cv._pack(w, 1.0), cv._pack(w.transpose(0, 1), 1.0)
    ref = F.conv2d(x.double().cpu(), w.double().cpu(), padding=1)
    check('conv s1 fp32', cv._conv_launch(x, wp, 128, 3, 3, 1, 1), ref)
    check('conv s1 split', cv._conv_launch(None, wp, 128, 3, 3, 1, 1, x_split=xs), ref)
    x2 = torch.randn(4, 256, 65, 65, device='cuda').contiguous(memory_format=torch.channels_last) * 1e-3
    x2s = sp.split_pack(x2)
    w2 = torch.randn(512, 256, 3, 3, device='cuda')
    ref = F.conv2d(x2.double().cpu(), w2.double().cpu(), stride=2)
    check('conv s2 fp32', cv._conv_launch(x2, cv._pack(w2, 1.0), 512, 3, 3, 2, 0), ref)
    check('conv s2 split', cv._conv_launch(None, cv._pack(w2, 1.0), 512, 3, 3, 2, 0, x_split=x2s), ref)
    w1 = torch.randn(128, 64, 1, 1, device='cuda')
    ref = F.conv2d(x.double().cpu(), w1.double().cpu())
    check('conv 1x1 split', cv._conv_launch(None, cv._pack(w1, 1.0), 128, 1, 1, 1, 0, x_split=xs), ref)
    g = torch.randn(2, 128, 32, 32, device='cuda').contiguous(memory_format=torch.channels_last) * 1e-5
    gs = sp.split_pack(g)
    ref = F.conv_transpose2d(g.double().cpu(), w.double().cpu(), padding=1)
    check('convT s1 fp32', cv._convT_launch(g, wpT, 64, 3, 3, 1, 1, (32, 32)), ref)
    check('convT s1 split', cv._convT_launch(None, wpT, 64, 3, 3, 1, 1, (32, 32), x_split=gs), ref)
    g2 = torch.randn(2, 128, 16, 16, device='cuda').contiguous(memory_format=torch.channels_last)
    g2s = sp.split_pack(g2)
    ref = F.conv_transpose2d(g2.double().cpu(), w.double().cpu(), stride=2)
    check('convT s2 (ct2) fp32', cv._convT_launch(g2, wpT, 64, 3, 3, 2, 0, (33, 33)), ref)
    check('convT s2 (ct2) split', cv._convT_launch(None, wpT, 64, 3, 3, 2, 0, (33, 33), x_split=g2s), ref)
    for ci, co, r in [(512, 512, 64), (256, 256, 128), (128, 128, 256), (512, 512, 32)]:
        x = torch.randn(B, ci, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
        w = torch.randn(co, ci, 3, 3, device='cuda')
        xs, wp = sp.split_pack(x), cv._pack(w, 1.0)
        flops = 2.0 * B * r * r * ci * co * 9
        t0 = timeit(lambda: cv._conv_launch(x, wp, co, 3, 3, 1, 1))
        t1 = timeit(lambda: cv._conv_launch(None, wp, co, 3, 3, 1, 1, x_split=xs))
        print(f'conv s1 {ci:4d}->{co:4d} @{r:3d} N{B}: fp32 {t0*1e6:7.1f} us {flops/t0/1e12:6.1f} TF | split {t1*1e6:7.1f} us {flops/t1/1e12:6.1f} TF ({t0/t1:.2f}x)')
    for ci, co, r in [(128, 256, 256), (256, 512, 128), (512, 512, 64)]:
        x = torch.randn(B, ci, r + 1, r + 1, device='cuda').contiguous(memory_format=torch.channels_last)
        w = torch.randn(co, ci, 3, 3, device='cuda')
        xs, wp, wpT = sp.split_pack(x), cv._pack(w, 1.0), cv._pack(w.transpose(0, 1), 1.0)
        flops = 2.0 * B * (r // 2) ** 2 * ci * co * 9
        t0 = timeit(lambda: cv._conv_launch(x, wp, co, 3, 3, 2, 0))
        t1 = timeit(lambda: cv._conv_launch(None, wp, co, 3, 3, 2, 0, x_split=xs))
        gy = torch.randn(B, co, r // 2, r // 2, device='cuda').contiguous(memory_format=torch.channels_last)
        gs = sp.split_pack(gy)
        t2 = timeit(lambda: cv._convT_launch(gy, wpT, ci, 3, 3, 2, 0, (r + 1, r + 1)))
        t3 = timeit(lambda: cv._convT_launch(None, wpT, ci, 3, 3, 2, 0, (r + 1, r + 1), x_split=gs))
        print(f'conv s2 {ci:4d}->{co:4d} @{r:3d} N{B}: fp32 {t0*1e6:7.1f} us {flops/t0/1e12:6.1f} TF | split {t1*1e6:7.1f} us {flops/t1/1e12:6.1f} TF ({t0/t1:.2f}x)'
              f' || convT2 fp32 {t2*1e6:7.1f} us {flops/t2/1e12:6.1f} TF | split {t3*1e6:7.1f} us {flops/t3/1e12:6.1f} TF ({t2/t3:.2f}x)')
