"""Long steady-state run of the trainer with step graphs: time per 100 iterations, losses, peak memory (GPU)."""
import os, random, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.models import Discriminator, Generator
from rick_amd.synth import synth_latents, synth_reals
from rick_amd.train import RickTrainer, TrainConfig
torch.manual_seed(1)
random.seed(int(os.environ.get('SEED', 1)))      # the mixing decisions (train.py: random.random / randint) — unseeded, every run takes its own trajectory
dev = 'cuda'
cfg = TrainConfig(batch=4)
g, d = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
ge, de = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
tr = RickTrainer(cfg, g, d, ge, de)
real = [synth_reals(4, 256, seed=s).to(dev) for s in range(4)]
tr.enable_graphs('--eager' not in sys.argv)
tr.prepare_graphs(real[0])
i0 = cfg.warmup_iter + 1
# FISHER=1: the full RICK loop — a Fisher sweep and new filter masks every cfg.fisher_freq iterations (pruned filters are the
# channel groups of zeros / tiny values that the conv kernels' operand exponent has to survive)
fisher_in = None
if os.environ.get('FISHER'):
    nf = cfg.num_fisher_img
    fisher_in = ([synth_latents(1, seed=500 + j).to(dev) for j in range(nf)], [synth_reals(1, 256, seed=600 + j).to(dev) for j in range(nf)])
    tr.fisher_sweep(*fisher_in, first=True)
    i0 = cfg.warmup_iter
for blk in range(int(os.environ.get('BLOCKS', 5))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(100):
        tr.iteration(i0 + blk * 100 + k, real[k % 4], fisher_in)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100 * 1e3
    L = {k: float(v) for k, v in tr.losses.items()}
    print(f'block {blk}: {dt:.2f} ms/iter  d={L["d"]:.4f} g={L["g"]:.4f} r1={L["r1"]:.5f} path={L["path"]:.5f} '
          f'mpl={float(tr.mean_path_length):.4f} mem={torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB', flush=True)
assert all(bool(torch.isfinite(p).all()) for p in g.parameters())
print('finite ok')
import ctypes
from rick_amd._lib import lib
c = ctypes.c_uint(0)
assert lib.rick_saturation_count(ctypes.byref(c), 0) == 0
assert c.value == 0, f'{c.value} fp16 saturation events: some operand sat far above its block\'s sampled maximum (finite, wrong products)'
print('no saturation events')
