"""Time of the one-launch weight re-pack of a network (rick_conv_pack_weights_multi) after an optimiser step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd import op
from rick_amd.models import Discriminator, Generator
from rick_amd.synth import synth_reals
from rick_amd.train import RickTrainer, TrainConfig
torch.manual_seed(1)
dev = 'cuda'
g, d = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
ge, de = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
tr = RickTrainer(TrainConfig(batch=4, num_fisher_img=1), g, d, ge, de)
real = synth_reals(4, 256, seed=1).to(dev)
tr.d_step(real, [torch.randn(4, 512, device=dev)])
tr.g_step([torch.randn(4, 512, device=dev)])
torch.cuda.synchronize()
for name, grp, flat in (('G', tr._pack_groups[0], tr.g_flat), ('D', tr._pack_groups[1], tr.d_flat)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(10):
        op.bump_weights_epoch(flat.params)
        e0.record(); grp.refresh(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f'{name}: {grp.n} packed views, {grp.total_blocks} blocks: {sorted(ts)[len(ts) // 2]:.1f} us')
