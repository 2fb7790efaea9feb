"""Kernel mix of ONE step type (eager issue; run under rocprofv3 --kernel-trace --stats): python tools/prof_one_step.py plr|r1|d|g"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.models import Discriminator, Generator
from rick_amd.synth import synth_reals
from rick_amd.train import RickTrainer, TrainConfig, mixing_noise
torch.manual_seed(1)
dev = 'cuda'
cfg = TrainConfig(batch=4, num_fisher_img=1)
g, d = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
ge, de = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
tr = RickTrainer(cfg, g, d, ge, de)
real = synth_reals(4, 256, seed=1).to(dev)
nz = lambda b: mixing_noise(b, 512, 0.9, dev)
which = sys.argv[1]
fn = {'d': lambda: tr.d_step(real, nz(4)), 'g': lambda: tr.g_step(nz(4)), 'r1': lambda: tr.r1_step(real),
      'plr': lambda: tr.plr_step(nz(2))}[which]
for _ in range(10):
    fn()
torch.cuda.synchronize()
