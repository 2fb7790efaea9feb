"""Per-step-type timing of the trainer (GPU): d_step, g_step, r1_step, plr_step, ema."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.models import Discriminator, Generator
from rick_amd.synth import synth_latents, synth_reals
from rick_amd.train import RickTrainer, TrainConfig, mixing_noise
torch.manual_seed(1)
cfg = TrainConfig(batch=4, num_fisher_img=1)
dev = 'cuda'
g, d = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
ge, de = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
tr = RickTrainer(cfg, g, d, ge, de)
real = synth_reals(4, 256, seed=1).to(dev)
tr.fisher_sweep([synth_latents(1, seed=5).to(dev)], [synth_reals(1, 256, seed=6).to(dev)], first=True)
def T(fn, reps=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
nz = lambda b: mixing_noise(b, 512, 0.9, dev)
print('d_step   %.2f ms' % T(lambda: tr.d_step(real, nz(4))))
print('g_step   %.2f ms' % T(lambda: tr.g_step(nz(4))))
print('r1_step  %.2f ms' % T(lambda: tr.r1_step(real)))
print('plr_step %.2f ms' % T(lambda: tr.plr_step(nz(2))))
print('ema      %.2f ms' % T(lambda: tr.ema_step()))
with torch.no_grad():
    print('G fwd B=4 (no grad) %.2f ms' % T(lambda: g(nz(4))))
    print('D fwd B=8 (no grad) %.2f ms' % T(lambda: d(torch.cat([real, real]), calls=2)))
print('fisher sample %.2f ms' % T(lambda: tr.fisher_sweep([synth_latents(1, seed=5).to(dev)], [synth_reals(1, 256, seed=6).to(dev)], first=True), reps=3))
