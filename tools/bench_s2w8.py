"""Stride-2 eight-wave 256 co x 128 position igemm blocks (conv.hip, igemm_body WDMA = 5) against the four-wave 128 x 64 blocks:
values (bit-equal on split images, fp64 error on fp32 operands) and launch time, both forms in ONE process, interleaved rounds.
GPU.  usage: [B=8] python tools/bench_s2w8.py [check] [time]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd._lib import lib
from rick_amd.op import conv as cv, split as sp

S2W8, MINBLK = 4, 1
B = int(os.environ.get('B', 8))
which = sys.argv[1:] or ['check', 'time']
F = torch.nn.functional
torch.manual_seed(0)


def nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())


if 'check' in which:
    lib.rick_conv_tuning(MINBLK, 1)
    for (n, ci, co, r) in [(2, 128, 256, 33), (1, 160, 256, 41), (3, 256, 512, 17), (8, 128, 256, 129), (8, 256, 512, 65)]:
        x = nhwc(torch.randn(n, ci, r, r, device='cuda') * torch.exp2(torch.randint(-6, 3, (n, ci, 1, 1), device='cuda').float()))
        wt = torch.randn(co, ci, 3, 3, device='cuda')
        so = torch.rand(n, co, device='cuda') + 0.5
        wp = cv._pack(wt, 1.0)
        ref = F.conv2d(x.double().cpu(), wt.double().cpu(), stride=2) * so.double().cpu()[:, :, None, None]
        out = {}
        for on in (0, 1):
            lib.rick_conv_tuning(S2W8, 2 * on)
            out[on] = (cv._conv_launch(x, wp, co, 3, 3, 2, 0, oscale=so),)
            if ci % 32 == 0:
                out[on] += (cv._conv_launch(None, wp, co, 3, 3, 2, 0, x_split=sp.split_pack(x)),)
        torch.cuda.synchronize()
        print(f'N{n} {ci}->{co} @{r}: fprop s2 vs fp64: w4 {rel(out[0][0], ref):.2e} w8 {rel(out[1][0], ref):.2e}'
              + (f' | split bit-equal {torch.equal(out[0][1], out[1][1])} (err {rel(out[1][1], ref / so.double().cpu()[:, :, None, None]):.2e})' if len(out[0]) > 1 else ''))
    lib.rick_conv_tuning(MINBLK, 192)

if 'time' in which:
    def run_rounds(fns, rounds=7, reps=10):
        ts = [[] for _ in fns]
        for f in fns:
            for _ in range(3):
                f()
        for _ in range(rounds):
            for i, f in enumerate(fns):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    f()
                e1.record()
                torch.cuda.synchronize()
                ts[i].append(e0.elapsed_time(e1) / reps * 1e-3)
        return [sorted(t)[len(t) // 2] for t in ts]

    print(f'B={B}: median over interleaved rounds; w4 = four-wave 128 co x 64 positions, w8 = eight-wave 256 co x 128 positions')
    lib.rick_conv_tuning(MINBLK, 64)
    for ci, co, r in [(128, 256, 256), (256, 512, 128), (512, 512, 64), (512, 512, 32)]:
        x = nhwc(torch.randn(B, ci, r + 1, r + 1, device='cuda'))
        wt = torch.randn(co, ci, 3, 3, device='cuda')
        wp = cv._pack(wt, 1.0)
        xs = sp.split_pack(x)
        flops = 2.0 * B * (r // 2) ** 2 * ci * co * 9

        def mk(on, fn):
            def f():
                lib.rick_conv_tuning(S2W8, 2 * on)
                fn()
            return f
        for name, fn in [('fp32', lambda: cv._conv_launch(x, wp, co, 3, 3, 2, 0)), ('split', lambda: cv._conv_launch(None, wp, co, 3, 3, 2, 0, x_split=xs))]:
            m4, m8 = run_rounds([mk(0, fn), mk(1, fn)])
            print(f's2 {ci:4d}->{co:4d} @{r:3d}->{r // 2:3d} {name:6s}: w4 {m4*1e6:7.1f} us {flops/m4/1e12:6.1f} TF | w8 {m8*1e6:7.1f} us {flops/m8/1e12:6.1f} TF | {m4/m8:.3f}x')
    lib.rick_conv_tuning(MINBLK, 192)
    lib.rick_conv_tuning(S2W8, 0)
