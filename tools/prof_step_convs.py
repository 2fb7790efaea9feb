"""Conv-family launches of one step type by layer tag (HIP events around every launch, eager issue):
python tools/prof_step_convs.py [d|g|r1|plr]"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.models import Discriminator, Generator
from rick_amd.op.conv import launch_profiler
from rick_amd.synth import synth_reals
from rick_amd.train import RickTrainer, TrainConfig
torch.manual_seed(1)
dev = 'cuda'
cfg = TrainConfig(batch=4, num_fisher_img=1)
g, d = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
ge, de = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
tr = RickTrainer(cfg, g, d, ge, de)
real = synth_reals(4, 256, seed=1).to(dev)
which = (sys.argv[1:] or ['plr'])[0]
tr.enable_graphs(False)
nz = lambda n: [torch.randn(n, 512, device=dev)]
fns = {'d': lambda: tr.d_step(real, nz(4)), 'r1': lambda: tr.r1_step(real),
       'g': lambda: tr.g_step(nz(4)), 'plr': lambda: tr.plr_step(nz(2))}
tr.d_step(real, nz(4))
fns[which]()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with launch_profiler() as prof:
    e0.record()
    fns[which]()
    e1.record()
    torch.cuda.synchronize()
by = {}
for kind, flops, a, b, tag, *_ in prof:
    t = by.setdefault(tag, [0.0, 0.0, 0])
    t[0] += flops; t[1] += a.elapsed_time(b) * 1e-3; t[2] += 1
tot = sum(v[1] for v in by.values())
print(f'== {which}: eager step {e0.elapsed_time(e1):.2f} ms, conv family {tot * 1e3:.2f} ms in {sum(v[2] for v in by.values())} launches')
for tag, (fl, dt, n) in sorted(by.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f'  {tag:46s} n={n:3d} {dt * 1e6:8.1f} us  {fl / dt / 1e12:6.1f} TF')
