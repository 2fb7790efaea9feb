import sys, os, torch
sys.path.insert(0, '/root/repo')
from rick_amd.op.upfirdn2d import upfirdn2d
from tools.bench_conv_util import timeit
k4 = torch.tensor([1., 3., 3., 1.], device='cuda'); k4 = (k4[:, None] * k4[None, :]) / 16
for c in (128, 256, 512):
    for w in (64, 65, 66, 72, 128, 129, 130, 136, 144, 257):
        for pad in ((1, 1), (2, 2)):
            x = torch.randn(4, c, w, w, device='cuda').contiguous(memory_format=torch.channels_last)
            t = timeit(lambda: upfirdn2d(x, k4, pad=pad))
            ow = w + 2 * pad[0] - 3
            print(f'C={c} in {w}x{w} pad{pad} out {ow}: {t*1e6:7.1f} us  {(x.numel()*4 + 4*c*ow*ow*4)/t/1e12:5.2f} TB/s')
