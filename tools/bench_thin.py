import sys, os, torch
sys.path.insert(0, '/root/repo')
from rick_amd.op import misc
from tools.bench_conv_util import timeit
for c, r in [(512, 64), (256, 128), (128, 256)]:
    x = torch.randn(4, c, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
    W = torch.randn(4, 3, c, device='cuda')
    t = timeit(lambda: misc.thin_fwd(x, W), reps=30)
    print(f'NQ={os.environ.get("RICK_THIN_NQ")} C={c} @{r}: {t*1e6:6.1f} us {x.numel()*4/t/1e12:5.2f} TB/s')
for c, r in [(512, 64), (256, 128), (128, 256)]:
    x = torch.randn(4, c, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
    tt = torch.randn(4, 3, r, r, device='cuda')
    W = torch.randn(4, 3, c, device='cuda')
    t = timeit(lambda: misc._ThinWgrad.apply(tt, x), reps=30)
    t2 = timeit(lambda: misc.thin_bwdx(tt, W), reps=30)
    print(f'C={c} @{r}: thin_wgrad {t*1e6:6.1f} us {x.numel()*4/t/1e12:5.2f} TB/s | thin_bwdx {t2*1e6:6.1f} us {x.numel()*4/t2/1e12:5.2f} TB/s (write)')

# device-side kernel durations (the event-timed loop above is host-bound below ~15 us per op: every <= 33 MB row reads 12-16 us)
print('--- kernel durations (torch.profiler, device time)')
from torch.profiler import ProfilerActivity, profile
for c, r in [(512, 64), (256, 128), (128, 256)]:
    x = torch.randn(4, c, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
    tt = torch.randn(4, 3, r, r, device='cuda')
    W = torch.randn(4, 3, c, device='cuda')
    for _ in range(3):
        misc._ThinWgrad.apply(tt, x); misc.thin_bwdx(tt, W); misc.thin_fwd(x, W)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(20):
            misc._ThinWgrad.apply(tt, x); misc.thin_bwdx(tt, W); misc.thin_fwd(x, W)
        torch.cuda.synchronize()
    mb = x.numel() * 4 / 1e6
    for ev in prof.key_averages():
        if 'thin' in ev.key:
            us = ev.device_time_total / ev.count
            print(f'C={c} @{r} ({mb:6.1f} MB): {ev.key[:60]:60s} {us:7.1f} us  {mb / us:5.2f} TB/s')
