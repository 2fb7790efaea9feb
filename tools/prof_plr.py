import sys, os, cProfile, pstats, io, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.models import Discriminator, Generator
from rick_amd.synth import synth_latents, synth_reals
from rick_amd.train import RickTrainer, TrainConfig, mixing_noise
torch.manual_seed(1)
dev = 'cuda'
cfg = TrainConfig(batch=4, num_fisher_img=1)
g, d = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
ge, de = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
tr = RickTrainer(cfg, g, d, ge, de)
nz = lambda b: mixing_noise(b, 512, 0.9, dev)
for _ in range(2): tr.plr_step(nz(2))
torch.cuda.synchronize()
# GPU-only time estimate: enqueue several and sync
t0 = time.perf_counter()
for _ in range(4): tr.plr_step(nz(2))
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('plr: host enqueue %.1f ms/step, after-sync extra %.1f ms total' % ((t1 - t0) / 4 * 1e3, (t2 - t1) * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(3): tr.plr_step(nz(2))
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(30); print(s.getvalue()[:7000])
