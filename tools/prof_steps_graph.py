"""GPU time of each step type when replayed from its captured graph (no host issue cost), and the kernel mix of the
path-length step (run under rocprofv3 --kernel-trace --stats)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.models import Discriminator, Generator
from rick_amd.synth import synth_reals
from rick_amd.train import RickTrainer, TrainConfig
torch.manual_seed(1)
dev = 'cuda'
cfg = TrainConfig(batch=4, num_fisher_img=1)
g, d = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
ge, de = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
tr = RickTrainer(cfg, g, d, ge, de)
real = synth_reals(4, 256, seed=1).to(dev)
tr.enable_graphs(True)
which = sys.argv[1:] or ['d', 'r1', 'g', 'plr']
tr._real = real.clone()
tr.d_step(tr._real, None, graph=True)          # one D step so that every network has been used once
fns = {'d': lambda: tr.d_step(tr._real, None, graph=True), 'r1': lambda: tr.r1_step(tr._real, graph=True),
       'g': lambda: tr.g_step(None, graph=True), 'plr': lambda: tr.plr_step(None, graph=True)}
for k in which:
    for _ in range(3): fns[k]()                  # two eager runs + the capture
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): fns[k]()
    torch.cuda.synchronize()
    print(f'{k:4s} graph replay {(time.perf_counter() - t0) / 8 * 1e3:7.2f} ms')
