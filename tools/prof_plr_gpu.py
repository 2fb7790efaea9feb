"""How much of the path-length / R1 steps is GPU work vs host time: wall time per step vs the sum of kernel
time measured with events around a fully queued run."""
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
nz = lambda b: mixing_noise(b, 512, 0.9, dev)
def T(fn, reps=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    t_host = (time.perf_counter() - t0) / reps * 1e3      # host time to ENQUEUE (no sync)
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / reps * 1e3
    return t_host, t_all
for name, fn in (('d_step', lambda: tr.d_step(real, nz(4))), ('g_step', lambda: tr.g_step(nz(4))),
                 ('r1_step', lambda: tr.r1_step(real)), ('plr_step', lambda: tr.plr_step(nz(2))), ('ema', lambda: tr.ema_step())):
    h, a = T(fn)
    print(f'{name:9s} host-enqueue {h:7.2f} ms   wall {a:7.2f} ms')
# a whole 16-iteration cycle
def cycle():
    for i in range(16):
        tr.iteration(256 + i, real)
h, a = T(cycle, reps=2)
print(f'16-iteration cycle: host-enqueue {h/16:.2f} ms/iter, wall {a/16:.2f} ms/iter')
