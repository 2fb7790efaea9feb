"""Micro-benchmark of the HBM-bound kernels at the layer shapes of the 256-px networks (GPU).
Prints device time and algorithmic GB/s (bytes that must move once / device time) per op and shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd import op                                    # noqa: E402
from rick_amd.op import fused_act, misc                    # noqa: E402
from rick_amd.op.upfirdn2d import upfirdn2d                # noqa: E402


def timeit(fn, reps=20, warm=3):
    """GPU time of one call = the summed DEVICE durations of the kernels it launches (torch.profiler), averaged over `reps`.
    (Round 4 timed a back-to-back loop with two events: every op below ~30 MB then read the HOST's 12-27 us per Python call,
    and a first-touch allocation inside the loop produced the 2 057 us `act_bwd` row of r04_hbm_microbench.txt.)"""
    from torch.profiler import ProfilerActivity, profile
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    return sum(ev.device_time_total for ev in prof.key_averages()) / reps * 1e-6


B = int(os.environ.get('B', 4))
shapes = [(512, 8), (512, 16), (512, 32), (512, 64), (256, 128), (128, 256)]
k4 = torch.tensor([1., 3., 3., 1.], device='cuda')
k4 = (k4[:, None] * k4[None, :]) / 64


def nhwc(*s):
    return torch.randn(*s, device='cuda').contiguous(memory_format=torch.channels_last)


for c, r in shapes:
    x = nhwc(B, c, r, r)
    g = nhwc(B, c, r, r)
    bias = torch.randn(c, device='cuda')
    noise = torch.randn(B, 1, r, r, device='cuda')
    nw = torch.zeros(1, device='cuda') + 0.1
    nb = x.numel() * 4
    out = [f'{c:4d}ch @{r:3d} ({nb/1e6:6.1f} MB)']
    t = timeit(lambda: fused_act.fused_leaky_relu(x, bias))
    out.append(f'act {t*1e6:6.1f}us {2*nb/t/1e9:6.0f}')
    t = timeit(lambda: fused_act.fused_noise_bias_act(x, bias, noise, nw))
    out.append(f'act+noise {t*1e6:6.1f}us {2*nb/t/1e9:6.0f}')
    y = fused_act.fused_noise_bias_act(x, bias, noise, nw)
    t = timeit(lambda: fused_act._ActAdjoint.apply(g, y, noise, 0.2, 2 ** 0.5, True, True))
    out.append(f'act_bwd {t*1e6:6.1f}us {3*nb/t/1e9:6.0f}')
    t = timeit(lambda: misc._hw_dot_raw(x, g))
    out.append(f'hw_dot {t*1e6:6.1f}us {2*nb/t/1e9:6.0f}')
    s = torch.rand(B, c, device='cuda')
    t = timeit(lambda: misc._chan_scale_raw(x, s))
    out.append(f'chan_scale {t*1e6:6.1f}us {2*nb/t/1e9:6.0f}')
    # blur after transposed conv: [B,C,r+1,r+1] -> [B,C,r,r], pad (1,1)
    xb = nhwc(B, c, r + 1, r + 1)
    t = timeit(lambda: upfirdn2d(xb, k4 * 4, pad=(1, 1)))
    out.append(f'blur11 {t*1e6:6.1f}us {(xb.numel()*4+nb)/t/1e9:6.0f}')
    # D: blur pad (2,2) r -> r+1 and the skip path down 2
    t = timeit(lambda: upfirdn2d(x, k4, pad=(2, 2)))
    out.append(f'blur22 {t*1e6:6.1f}us {(nb + B*c*(r+1)**2*4)/t/1e9:6.0f}')
    t = timeit(lambda: upfirdn2d(x, k4, down=2, pad=(1, 1)))
    out.append(f'down2 {t*1e6:6.1f}us {(nb + nb/4)/t/1e9:6.0f}')
    # ToRGB thin ops
    W = torch.randn(B, 3, c, device='cuda')
    t = timeit(lambda: misc.thin_fwd(x, W))
    out.append(f'thin_fwd {t*1e6:6.1f}us {nb/t/1e9:6.0f}')
    print(' | '.join(out), flush=True)
