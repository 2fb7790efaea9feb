"""One kernel, one shape, a few launches — target for `rocprofv3 --pmc ...` (put `python3 tools/pmc_kernel.py ...`
directly after `--`).  usage: pmc_kernel.py <conv|convT2|convT2_old|conv_s2|wgrad|wgrad_s2> ci co r [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.op import conv as cv  # noqa: E402

mode, ci, co, r = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 4
SPLIT = mode.endswith('_split')          # operands as split images (rick_amd/op/split.py)
mode = mode[:-6] if SPLIT else mode
from rick_amd.op import split as sp  # noqa: E402
w = torch.randn(co, ci, 3, 3, device='cuda')
wp = cv._pack(w, 1.0)
if mode in ('conv', 'wgrad'):
    x = torch.randn(B, ci, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
    fn = (lambda: cv._conv_launch(x, wp, co, 3, 3, 1, 1)) if mode == 'conv' else (lambda: cv._wgrad_launch(gy, x, 3, 3, 1, 1))
    if SPLIT:
        xs, gs = sp.split_pack(x), sp.split_pack(gy)
        fn = ((lambda: cv._conv_launch(None, wp, co, 3, 3, 1, 1, x_split=xs)) if mode == 'conv'
              else (lambda: cv._wgrad_launch(None, None, 3, 3, 1, 1, a_split=gs, b_split=xs)))
elif mode in ('conv_s2', 'wgrad_s2'):
    x = torch.randn(B, ci, 2 * r + 1, 2 * r + 1, device='cuda').contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
    fn = (lambda: cv._conv_launch(x, wp, co, 3, 3, 2, 0)) if mode == 'conv_s2' else (lambda: cv._wgrad_launch(gy, x, 3, 3, 2, 0))
    if SPLIT:
        xs, gs = sp.split_pack(x), sp.split_pack(gy)
        fn = ((lambda: cv._conv_launch(None, wp, co, 3, 3, 2, 0, x_split=xs)) if mode == 'conv_s2'
              else (lambda: cv._wgrad_launch(None, None, 3, 3, 2, 0, a_split=gs, b_split=xs)))
else:   # transposed stride 2: input [B, ci, r, r] -> [B, co, 2r+1, 2r+1]
    cv._USE_CT2 = mode == 'convT2'
    x = torch.randn(B, ci, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
    fn = lambda: cv._convT_launch(x, wp, co, 3, 3, 2, 0, (2 * r + 1, 2 * r + 1))   # noqa: E731
    if SPLIT:
        xs = sp.split_pack(x)
        fn = lambda: cv._convT_launch(None, wp, co, 3, 3, 2, 0, (2 * r + 1, 2 * r + 1), x_split=xs)   # noqa: E731
for _ in range(6):
    fn()
torch.cuda.synchronize()
