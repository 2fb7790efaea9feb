"""Where do the D2D copies / torch elementwise launches of the first-order steps come from?  One eager D step and one eager G step
under torch.profiler with stacks; prints, per aten op that launches a copy / add / mul / fill kernel, the call count and the
innermost rick_amd source line."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.models import Discriminator, Generator
from rick_amd.synth import synth_reals
from rick_amd.train import RickTrainer, TrainConfig
from torch.profiler import ProfilerActivity, profile

torch.manual_seed(1)
dev = 'cuda'
g, d = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
tr = RickTrainer(TrainConfig(batch=4), g, d, Generator(256, 512, 8).to(dev), Discriminator(256).to(dev))
real = synth_reals(4, 256, seed=1).to(dev)
tr.enable_graphs(True)
which = sys.argv[1] if len(sys.argv) > 1 else 'd'
fn = {'d': lambda: tr.d_step(real, None, graph=True), 'g': lambda: tr.g_step(None, graph=True)}[which]
tr.d_step(real, None, graph=True)       # eager warm-ups (n <= 2 run eagerly)
tr.g_step(None, graph=True)
tr._gs[which]['n'] = 0                  # stay eager
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    fn()
    torch.cuda.synchronize()
rows = collections.Counter()
for ev in prof.events():
    if ev.device_type.name == 'CPU' and ev.name in ('aten::copy_', 'aten::add', 'aten::add_', 'aten::mul', 'aten::mul_', 'aten::fill_', 'aten::zero_',
                                                   'aten::clone', 'aten::contiguous', 'aten::cat', 'aten::sum', 'aten::div', 'aten::sub', 'aten::neg', 'aten::index_select', 'aten::where'):
        frame = next((s for s in ev.stack if 'rick_amd' in s), ev.stack[0] if ev.stack else '?')
        shp = str(ev.input_shapes[:2])[:60]
        rows[(ev.name, frame.split('/root/repo/')[-1][:90], shp)] += 1
for (name, frame, shp), n in sorted(rows.items(), key=lambda kv: -kv[1])[:60]:
    print(f'{n:4d} x {name:18s} {shp:60s} {frame}')
print('kernels:')
for ev in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:0]:
    pass
