"""convt2 block quantisation: time the one-pass transposed conv over batch sizes and print blocks / 256 next to TFLOP/s.
usage (GPU): python tools/ct2_rounds.py"""
import sys, os, ctypes, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.op import conv as cv
from rick_amd._lib import lib
from tools.bench_conv_util import timeit

SHAPES = [(512, 256, 64), (256, 128, 128), (512, 512, 32), (512, 512, 16)]
BATCHES = (1, 2, 3, 4, 6, 7, 8)
if os.environ.get('CT2_B'):
    BATCHES = tuple(int(b) for b in os.environ['CT2_B'].split(','))
if os.environ.get('CT2_ONLY'):          # "ci,co,ih,B": one case (with RICK_HIP_LIB=...abl.so RICK_CT2_TILE / RICK_CT2_SUBQ overrides)
    ci, co, ih, b = map(int, os.environ['CT2_ONLY'].split(','))
    SHAPES, BATCHES = [(ci, co, ih)], (b,)
for ci, co, ih in SHAPES:
    wp = cv._pack(torch.randn(co, ci, 3, 3, device='cuda'), 1.0)
    for B in BATCHES:
        x = torch.randn(B, ci, ih, ih, device='cuda').contiguous(memory_format=torch.channels_last)
        out8 = (ctypes.c_int * 8)()
        lib.rick_convt2_plan(B, ih, ih, ci, co, 2 * ih + 1, 2 * ih + 1, out8)
        tw, th, nb, tiles, ns, cps, nfull, subq = list(out8)
        flops = 2.0 * B * ih * ih * ci * co * 9
        t = timeit(lambda: cv._convT_launch(x, wp, co, 3, 3, 2, 0, (2 * ih + 1, 2 * ih + 1)))
        items = tiles * ns
        print(f'{ci}->{co} in {ih} B={B}: tile {tw}x{th}x{nb} items {items:5d} = {items/256:5.2f} rounds, split {ns}, '
              f'{nfull} whole + {items - nfull} x {subq}, {t*1e6:7.1f} us {flops/t/1e12:6.1f} TF')
