"""Which autograd adds / fills / copies does one eager step launch?  Groups aten ops by (name, input shapes) for one step
type.  usage: prof_adds.py <d|g|r1|plr>"""
import sys, os, collections
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
which = sys.argv[1]
tr._real = real
mk = lambda b=4: mixing_noise(b, 512, cfg.mixing, dev)
fns = {'d': lambda: tr.d_step(real, mk()), 'r1': lambda: tr.r1_step(real),
       'g': lambda: tr.g_step(mk()), 'plr': lambda: tr.plr_step(mk(max(1, 4 // cfg.path_batch_shrink)))}
for _ in range(2):
    tr.d_step(real, mk())
    fns[which]()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                            record_shapes=True) as prof:
    fns[which]()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.key_averages(group_by_input_shape=True):
    if ev.key.startswith('aten::') and ev.device_time_total > 0 and any(
            k in ev.key for k in ('add', 'fill', 'copy', 'zero', 'mul', 'sum', 'cat', 'stack', 'clone', 'neg', 'div')):
        agg[(ev.key, str(ev.input_shapes)[:110])][0] += ev.count
        agg[(ev.key, str(ev.input_shapes)[:110])][1] += ev.self_device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print(f'{which}: {sum(v[0] for _, v in rows)} pointwise aten launches, {tot / 1e3:.3f} ms device self time')
for (k, sh), (n, t) in rows[:45]:
    print(f'{t:9.1f} us {n:4d}x  {k:22s} {sh}')
if os.environ.get('BY_NAME'):
    byname = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.key_averages():
        if ev.self_device_time_total > 0:
            byname[ev.key][0] += ev.count
            byname[ev.key][1] += ev.self_device_time_total
    rows = sorted(byname.items(), key=lambda kv: -kv[1][1])
    print(f'--- by name: {sum(v[0] for _, v in rows)} events with device time, {sum(v[1] for _, v in rows) / 1e3:.3f} ms')
    for k, (n, t) in rows[:70]:
        print(f'{t:9.1f} us {n:5d}x  {k[:120]}')
if os.environ.get('STACKS'):
    # who launches the small copies / fills?  (python frames of the ops named in STACKS, comma separated)
    want = os.environ['STACKS'].split(',')
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                                record_shapes=True, with_stack=True) as prof2:
        fns[which]()
        torch.cuda.synchronize()
    seen = collections.Counter()
    for ev in prof2.events():
        if ev.name in want and ev.device_time_total > 0:
            frames = [f for f in (ev.stack or []) if 'rick_amd' in f or 'tools/' in f][:3]
            seen[(ev.name, str(ev.input_shapes)[:60], ' <- '.join(f.split('/')[-1][:60] for f in frames))] += 1
    for (name, shp, st), n in seen.most_common(40):
        print(f'{n:4d}x {name:14s} {shp:60s} {st}')
