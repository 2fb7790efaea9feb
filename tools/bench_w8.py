"""Eight-wave 128 co x 256 position igemm blocks (conv.hip, igemm_body NW = 8) against the four-wave 128 x 128 blocks: results
(bit-equal on split images, 2e-6 of fp64 on fp32 operands) and launch time, both forms in ONE process, interleaved rounds
(rick_conv_tuning switches the form per launch).  GPU.  usage: [B=8] python tools/bench_w8.py [check] [time]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd._lib import lib
from rick_amd.op import conv as cv, split as sp

W8, MINBLK = 0, 1
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
    # (the last two shapes fill the chip with four-wave blocks, so that form runs without split-K: same summation order)
    for (n, ci, co, h, w) in [(2, 128, 192, 32, 32), (1, 160, 128, 40, 24), (3, 256, 128, 16, 48), (4, 128, 128, 128, 128), (8, 512, 256, 64, 64)]:
        x = nhwc(torch.randn(n, ci, h, w, device='cuda') * torch.exp2(torch.randint(-6, 3, (n, ci, 1, 1), device='cuda').float()))
        wt = torch.randn(co, ci, 3, 3, device='cuda')
        si, so = torch.rand(n, ci, device='cuda') + 0.5, torch.rand(n, co, device='cuda') + 0.5
        wp, wpT = cv._pack(wt, 1.0), cv._pack(wt.transpose(0, 1), 1.0)
        ref = F.conv2d(x.double().cpu() * si.double().cpu()[:, :, None, None], wt.double().cpu(), padding=1) * so.double().cpu()[:, :, None, None]
        gy = nhwc(torch.randn(n, co, h, w, device='cuda') * 1e-4)
        refT = F.conv_transpose2d(gy.double().cpu(), wt.double().cpu(), padding=1)
        out = {}
        for on in (0, 1):
            lib.rick_conv_tuning(W8, 2 * on)
            out[on] = (cv._conv_launch(x, wp, co, 3, 3, 1, 1, iscale=si, oscale=so), cv._conv_launch(x, wp, co, 3, 3, 1, 1),
                       cv._convT_launch(gy, wpT, ci, 3, 3, 1, 1, (h, w)))
            if ci % 32 == 0:
                xs, gs = sp.split_pack(x), sp.split_pack(gy)
                out[on] += (cv._conv_launch(None, wp, co, 3, 3, 1, 1, x_split=xs),)
                if co % 32 == 0:
                    out[on] += (cv._convT_launch(None, wpT, ci, 3, 3, 1, 1, (h, w), x_split=gs),)
        torch.cuda.synchronize()
        ref0 = F.conv2d(x.double().cpu(), wt.double().cpu(), padding=1)
        print(f'N{n} {ci}->{co} @{h}x{w}: scaled fprop vs fp64: w4 {rel(out[0][0], ref):.2e} w8 {rel(out[1][0], ref):.2e} | plain '
              f'{rel(out[0][1], ref0):.2e} {rel(out[1][1], ref0):.2e} | dgrad {rel(out[0][2], refT):.2e} {rel(out[1][2], refT):.2e}'
              + (f' | split fprop bit-equal {torch.equal(out[0][3], out[1][3])}' if len(out[0]) > 3 else '')
              + (f' dgrad bit-equal {torch.equal(out[0][4], out[1][4])}' if len(out[0]) > 4 else ''))
        sat = torch.zeros(1, dtype=torch.int32)
    lib.rick_conv_tuning(MINBLK, 192)

if 'time' in which:
    def run_rounds(fns, rounds=7, reps=10):
        """interleaved rounds: per variant the median of `rounds` event-timed batches of `reps` launches"""
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
        return [sorted(t)[len(t) // 2] for t in ts], [min(t) for t in ts]

    SC = int(os.environ.get('SCALES', 0))
    print(f'B={B} scales={SC}: median (min) over interleaved rounds')
    for ci, co, r in [(512, 512, 64), (256, 256, 128), (128, 128, 256), (512, 256, 64), (256, 128, 128)]:
        x = nhwc(torch.randn(B, ci, r, r, device='cuda'))
        wt = torch.randn(co, ci, 3, 3, device='cuda')
        wp, wpT = cv._pack(wt, 1.0), cv._pack(wt.transpose(0, 1), 1.0)
        gy = nhwc(torch.randn(B, co, r, r, device='cuda'))
        si = torch.rand(B, ci, device='cuda') + 0.5 if SC else None
        so = torch.rand(B, co, device='cuda') + 0.5 if SC else None
        xs, gs = sp.split_pack(x), sp.split_pack(gy)
        flops = 2.0 * B * r * r * ci * co * 9

        def mk(on, fn):
            def f():
                lib.rick_conv_tuning(W8, 2 * on)
                fn()
            return f
        rows = [('fprop fp32', lambda: cv._conv_launch(x, wp, co, 3, 3, 1, 1, iscale=si, oscale=so)),
                ('fprop split', lambda: cv._conv_launch(None, wp, co, 3, 3, 1, 1, oscale=so, x_split=xs)),
                ('dgrad fp32', lambda: cv._convT_launch(gy, wpT, ci, 3, 3, 1, 1, (r, r), iscale=so, oscale=si)),
                ('dgrad split', lambda: cv._convT_launch(None, wpT, ci, 3, 3, 1, 1, (r, r), oscale=si, x_split=gs))]
        for name, fn in rows:
            (m4, m8), (n4, n8) = run_rounds([mk(0, fn), mk(1, fn)])
            print(f'{ci:4d}->{co:4d} @{r:3d} {name:12s}: w4 {m4*1e6:7.1f} us {flops/m4/1e12:6.1f} TF ({flops/n4/1e12:6.1f}) | '
                  f'w8 {m8*1e6:7.1f} us {flops/m8/1e12:6.1f} TF ({flops/n8/1e12:6.1f}) | {m4/m8:.3f}x')
    lib.rick_conv_tuning(W8, 1)
