"""Producer kernels of the split-image path in isolation (GPU): FIR variants, activation adjoint, merge, stand-alone pack."""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.models import make_kernel
from rick_amd.op import dblock, split as sp
from rick_amd.op.fused_act import _ActAdjoint
from rick_amd.op.upfirdn2d import _fir, _flipped
from rick_amd.op.misc import add_scale
from tools.bench_conv_util import timeit

taps = make_kernel([1, 3, 3, 1]).cuda()
flip = _flipped(taps)
B = int(os.environ.get('B', 8))
for C, H in [(128, 256), (256, 128), (512, 64)]:
    x = torch.randn(B, C, H, H, device='cuda').contiguous(memory_format=torch.channels_last)
    A = sp.amax(x)
    mb = x.numel() * 4 / 1e6
    word = sp.new_amax('cuda')
    rows = [('blur fp32 (plain kernel)', lambda: _fir(x, taps, (1, 1), (1, 1), (2, 2, 2, 2)), 2),
            ('blur -> split only', lambda: dblock._fir_ex(x, taps, 1, 1, (2, 2, 2, 2), split_bound=A, no_f32=True), 2),
            ('blur -> fp32 (XO, nothing else)', lambda: dblock._fir_ex(x, taps, 1, 1, (2, 2, 2, 2)), 2),
            ('blur -> fp32 + amax', lambda: dblock._fir_ex(x, taps, 1, 1, (2, 2, 2, 2), amax=word), 2),
            ('blur -> fp32 + split', lambda: dblock._fir_ex(x, taps, 1, 1, (2, 2, 2, 2), split_bound=A), 3),
            ('down2 fp32 (plain)', lambda: _fir(x, taps, (1, 1), (2, 2), (1, 1, 1, 1)), 1.25),
            ('down2 -> split only', lambda: dblock._fir_ex(x, taps, 1, 2, (1, 1, 1, 1), split_bound=A, no_f32=True), 1.25),
            ('amax pass', lambda: sp.amax(x, word), 1),
            ('stand-alone pack (given bound)', lambda: sp.split_pack(x, A), 2),
            ]
    y = torch.randn_like(x)
    g = torch.randn_like(x)
    rows += [('act adjoint fp32', lambda: _ActAdjoint.apply(g, y, None, 0.2, 1.0, True, False), 3),
             ('act adjoint -> 1 image', lambda: dblock._act_adjoint_split(g, y, 0.2, 1.0, A, None, True, None), 3),
             ('act adjoint -> 2 images', lambda: dblock._act_adjoint_split(g, y, 0.2, 1.0, A, 0.7, True, None), 4),
             ('merge fp32', lambda: add_scale(x, y, 0.7), 3)]
    h2 = H // 2
    gs = torch.randn(B, C, h2, h2, device='cuda').contiguous(memory_format=torch.channels_last)
    base = torch.randn_like(x)
    rows += [('up2 adjoint fp32 (plain)', lambda: _fir(gs, flip, (2, 2), (1, 1), (2, 1, 2, 1)), 1.25),
             ('up2 adjoint accumulate + amax', lambda: dblock._fir_ex(gs, flip, 2, 1, (2, 1, 2, 1), out=base, amax=word, accumulate=True), 2.25)]
    print(f'--- C={C} H={H} B={B}: tensor {mb:.0f} MB')
    for name, fn, passes in rows:
        t = timeit(fn, reps=20)
        print(f'{name:34s} {t * 1e6:8.1f} us  {passes * mb / 1e6 / t:6.2f} TB/s')
