"""Which lines of rick_amd/ launch torch's own kernels in one step type (eager run under a TorchDispatchMode, aten ops grouped by the
innermost rick_amd frame; custom HIP launches go through ctypes and do not show).  Usage: python tools/trace_torch_ops.py [d|g|r1|plr]"""
import sys, os, collections
import torch
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
which = (sys.argv[1:] or ['d'])[0]
tr.enable_graphs(True)      # the first two runs of a graphed step type are eager runs of exactly what gets captured
fns = {'d': lambda: tr.d_step(real, None, graph=True), 'r1': lambda: tr.r1_step(real, graph=True),
       'g': lambda: tr.g_step(None, graph=True), 'plr': lambda: tr.plr_step(None, graph=True)}
if which != 'd':
    tr.d_step(real, None, graph=True)
fns[which]()
torch.cuda.synchronize()

VIEWS = {'view', '_unsafe_view', 'reshape', '_reshape_alias', 'as_strided', 't', 'transpose', 'permute', 'expand', 'slice',
         'select', 'unsqueeze', 'squeeze', 'detach', 'alias', 'unbind', 'split', 'split_with_sizes', 'chunk', 'empty',
         'empty_like', 'empty_strided', 'new_empty', 'new_empty_strided', 'unfold', 'narrow', 'view_as', 'lift_fresh',
         'is_same_size', 'sym_size', 'stride', 'size', 'result_type', '_local_scalar_dense', 'flatten', 'unflatten',
         'is_nonzero', 'item', 'numel', 'dim', 'is_pinned', 'set_', 'resize_', 'record_stream', 'contiguous'}
by = collections.Counter()
names = collections.defaultdict(collections.Counter)


class Tracer(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split('.')[0]
        if name not in VIEWS:
            frame = 'autograd engine (no rick_amd frame)'
            for fs in reversed(traceback.extract_stack()[:-1]):
                if '/rick_amd/' in fs.filename:
                    frame = f"{fs.filename.split('/rick_amd/')[-1]}:{fs.lineno} {fs.name}: {fs.line[:60]}"
                    break
            by[frame] += 1
            names[frame][name] += 1
        return func(*args, **(kwargs or {}))


with Tracer():
    fns[which]()
    torch.cuda.synchronize()
print(f'== {which}: {sum(by.values())} aten ops that are not views / allocations')
for f, n in by.most_common(80):
    print(f'{n:5d}  {f:110s} ' + ', '.join(f'{k}x{v}' for k, v in names[f].most_common(4)))
