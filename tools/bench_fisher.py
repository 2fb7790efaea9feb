"""Fisher sweep (train_dynamic_update_prune.py:214-393; BASELINE config 5) wall time per sample on one GPU:
python tools/bench_fisher.py [samples] [--eager]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.models import Discriminator, Generator
from rick_amd.synth import synth_latents, synth_reals
from rick_amd.train import RickTrainer, TrainConfig
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
torch.manual_seed(1)
dev = 'cuda'
cfg = TrainConfig(batch=4, num_fisher_img=n)
g, d = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
ge, de = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
tr = RickTrainer(cfg, g, d, ge, de)
tr.enable_graphs('--eager' not in sys.argv)
lat = [synth_latents(1, seed=500 + j).to(dev) for j in range(n)]
real = [synth_reals(1, 256, seed=600 + j).to(dev) for j in range(n)]
tr.fisher_sweep(lat[:2], real[:2], first=True)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    tr.fisher_sweep(lat, real, first=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'fisher sweep, {n} samples (batch 1 each) + per-filter reduce + percentile decisions + mask upload: {dt * 1e3:.1f} ms '
          f'= {dt / n * 1e3:.1f} ms/sample')
