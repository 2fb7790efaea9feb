"""Device durations of the small-layer conv launches (torch.profiler): 512 -> 512 3x3 stride 1, its data gradient form and the
stride-2 transposed form at 4^2 ... 32^2, batch B (env), with the weight-streaming bound next to them:
packed weights 9.4 MB (fp16 hi + lo) must cross HBM once per launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.op import conv as cv
from torch.profiler import ProfilerActivity, profile

reps = int(os.environ.get('REPS', 20))
for B in (4, 8):
    for r in (4, 8, 16, 32):
        x = torch.randn(B, 512, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
        w = torch.randn(512, 512, 3, 3, device='cuda')
        wp = cv._pack(w, 1.0)
        wpT = cv._pack(w.transpose(0, 1), 1.0)
        fns = {'conv s1': lambda: cv._conv_launch(x, wp, 512, 3, 3, 1, 1),
               'convT s2 (up)': lambda: cv._convT_launch(x, wpT, 512, 3, 3, 2, 0, (2 * r + 1, 2 * r + 1)),
               'conv s2 (down)': lambda: cv._conv_launch(x, wp, 512, 3, 3, 2, 0) if r >= 8 else None}
        for name, fn in fns.items():
            if fn() is None:
                continue
            torch.cuda.synchronize()
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
            evs = [(ev.key.split('(')[0][-48:], ev.device_time_total / reps) for ev in prof.key_averages()]
            tot = sum(t for _, t in evs)
            print(f'B={B} {r:2d}^2 {name:15s} total {tot:6.1f} us : ' + ' + '.join(f'{k} {t:.1f}' for k, t in evs), flush=True)
