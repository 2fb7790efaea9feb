"""Micro-benchmark of the conv-family kernels on the layer shapes of the 256-px networks (GPU).
Prints algorithmic TFLOP/s (issued = 3x for bf16x3) per launch type."""
import sys, os, ctypes, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.op import conv as cv

def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

B = int(os.environ.get('B', 4))
DATA = os.environ.get('DATA', 'randn')   # 'sign': +-1 everywhere (no fp16-subnormal lo parts, minimal mantissa toggling)
_randn = torch.randn
if DATA == 'sign':
    torch.randn = lambda *a, **k: torch.sign(_randn(*a, **k))
if DATA == 'unif':      # magnitudes in [0.5, 1): full mantissa activity, lo parts in the fp16 NORMAL range
    torch.randn = lambda *a, **k: torch.sign(_randn(*a, **k)) * (0.5 + 0.5 * torch.rand(*a, **k))
if DATA == 'small':     # 90 % of the values 2^-8 of the rest: most lo parts are fp16 subnormals
    torch.randn = lambda *a, **k: _randn(*a, **k) * torch.where(torch.rand(*a, **k) < 0.9, 2.0 ** -8, 1.0)
SC = int(os.environ.get('SCALES', 0))   # 1: per-(image, channel) input/output scales as in the modulated convs
which = sys.argv[1:] or ['fprop', 'dgrad', 'wgrad']
shapes = [(512, 512, 4), (512, 512, 8), (512, 512, 16), (512, 512, 32), (512, 512, 64), (256, 256, 128), (128, 128, 256),
          (512, 256, 64), (256, 128, 128)]
print(f'B={B} precision={cv.get_precision()} scales={SC}')
for ci, co, r in shapes:
    x = torch.randn(B, ci, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 3, 3, device='cuda')
    gy = torch.randn(B, co, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
    flops = 2.0 * B * r * r * ci * co * 9
    wp = cv._pack(w, 1.0)
    wpT = cv._pack(w.transpose(0, 1), 1.0)
    si = torch.rand(B, ci, device='cuda') + 0.5 if SC else None
    so = torch.rand(B, co, device='cuda') + 0.5 if SC else None
    out = [f'{ci:4d}->{co:4d} @{r:3d}: {flops/1e9:7.2f} GF']
    if 'fprop' in which:
        t = timeit(lambda: cv._conv_launch(x, wp, co, 3, 3, 1, 1, iscale=si, oscale=so))
        out.append(f'fprop {t*1e6:8.1f} us {flops/t/1e12:6.1f} TF')
    if 'dgrad' in which:
        t = timeit(lambda: cv._convT_launch(gy, wpT, ci, 3, 3, 1, 1, (r, r), iscale=so, oscale=si))
        out.append(f'dgrad {t*1e6:8.1f} us {flops/t/1e12:6.1f} TF')
    if 'wgrad' in which:
        t = timeit(lambda: cv._wgrad_launch(gy, x, 3, 3, 1, 1, ascale=so, bscale=si))
        out.append(f'wgrad {t*1e6:8.1f} us {flops/t/1e12:6.1f} TF')
    print(' | '.join(out))
# stride-2 (D conv2) and transposed (G up): dedicated single-staging kernel (csrc/convt2.hip) vs the generic multi-class launch
for ci, co, r in [(128, 256, 256), (256, 512, 128), (512, 512, 64), (512, 512, 32), (512, 512, 16), (512, 512, 8)]:
    x = torch.randn(B, ci, r + 1, r + 1, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 3, 3, device='cuda')
    wp = cv._pack(w, 1.0)
    flops = 2.0 * B * (r // 2) ** 2 * ci * co * 9
    out = [f's2 {ci:4d}->{co:4d} @{r:3d}->{r//2}: {flops/1e9:7.2f} GF']
    if 'fprop' in which:
        t = timeit(lambda: cv._conv_launch(x, wp, co, 3, 3, 2, 0))
        out.append(f'fprop {t*1e6:8.1f} us {flops/t/1e12:6.1f} TF')
    gy = torch.randn(B, co, r // 2, r // 2, device='cuda').contiguous(memory_format=torch.channels_last)
    if 'wgrad' in which:
        t3 = timeit(lambda: cv._wgrad_launch(gy, x, 3, 3, 2, 0))
        out.append(f'wgrad {t3*1e6:8.1f} us {flops/t3/1e12:6.1f} TF')
    if 'dgrad' in which:
        wpT = cv._pack(w.transpose(0, 1), 1.0)
        so = torch.rand(B, co, device='cuda') + 0.5 if SC else None
        si = torch.rand(B, ci, device='cuda') + 0.5 if SC else None
        for flag in (True, False):
            cv._USE_CT2 = flag
            t2 = timeit(lambda: cv._convT_launch(gy, wpT, ci, 3, 3, 2, 0, (r + 1, r + 1), iscale=so, oscale=si))
            out.append(f'convT[{"ct2" if flag else "multi"}] {t2*1e6:8.1f} us {flops/t2/1e12:6.1f} TF')
        cv._USE_CT2 = True
    print(' | '.join(out))
