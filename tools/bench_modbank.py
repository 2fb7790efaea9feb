import sys, torch
sys.path.insert(0, '/root/repo')
from rick_amd.models import Generator
from tools.bench_conv_util import timeit
g = Generator(256, 512, 8).cuda()
lat = torch.randn(4, g.n_latent, 512, device='cuda')
bank = g._modulation_bank()
with torch.no_grad():
    sb = bank(lat)
    t = timeit(lambda: bank(lat), reps=50)
    db = g._demod_bank()
    t2 = timeit(lambda: db(sb), reps=50)
print(f'modbank fwd {t*1e6:.1f} us   demod bank fwd {t2*1e6:.1f} us')
with torch.no_grad(), torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
    for _ in range(20):
        sb = bank(lat)
        db(sb)
    torch.cuda.synchronize()
for ev in prof.key_averages():
    if 'kernel' in ev.key:
        print(f'{ev.key[:50]:50s} {ev.count:3d} x {ev.device_time_total / ev.count:6.1f} us')
# backward kernels of the banks (one G step's worth)
for p_ in g.style.parameters():
    p_.requires_grad_(False)
from rick_amd import op
for p_ in g.parameters():
    p_.grad = torch.zeros_like(p_)
noise = [torch.randn(1, 1, n.shape[-1], n.shape[-1], device='cuda') for n in g.make_noise()]
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        with op.grad_sink():
            img, _ = g([lat], input_is_latent=True, noise=noise)
            img.square().mean().backward()
    torch.cuda.synchronize()
for ev in prof.key_averages():
    if any(k in ev.key for k in ('demod', 'modbank', 'wsq')):
        print(f'{ev.key[:50]:50s} {ev.count:3d} x {ev.device_time_total / ev.count:6.1f} us')
