"""Run ONE step type repeatedly (eager) — target for rocprofv3 --kernel-trace --stats.  usage: prof_one_step.py <d|g|r1|plr> [reps]"""
import sys, os
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
which = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tr.enable_graphs(True)
tr._real = real
fns = {'d': lambda: tr.d_step(real, None, graph=True), 'r1': lambda: tr.r1_step(real, graph=True),
       'g': lambda: tr.g_step(None, graph=True), 'plr': lambda: tr.plr_step(None, graph=True)}
for _ in range(3):
    tr.d_step(real, None, graph=True)
    fns[which]()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fns[which]()
e1.record()
torch.cuda.synchronize()
print(f'{which}: {e0.elapsed_time(e1) / reps:.3f} ms per step (graph replay)')
