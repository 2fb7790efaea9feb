"""Which Python call sites launch device copies / fills / small torch ops in one eager step?  Patches a few Tensor methods and
functions with counters keyed by the first rick_amd frame.  usage: find_copies.py <d|g>"""
import sys, os, collections, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.models import Discriminator, Generator
from rick_amd.synth import synth_reals
from rick_amd.train import RickTrainer, TrainConfig, mixing_noise
torch.manual_seed(1)
cfg = TrainConfig(batch=4, num_fisher_img=1)
g, d = Generator(256, 512, 8).cuda(), Discriminator(256).cuda()
ge, de = Generator(256, 512, 8).cuda(), Discriminator(256).cuda()
tr = RickTrainer(cfg, g, d, ge, de)
real = synth_reals(4, 256, seed=1).cuda()
tr.enable_graphs(True)
tr._real = real
which = sys.argv[1]
fns = {'d': lambda: tr.d_step(real, None, graph=False) if False else tr.d_step(real, mixing_noise(4, 512, cfg.mixing, 'cuda')),
       'g': lambda: tr.g_step(mixing_noise(4, 512, cfg.mixing, 'cuda'))}
for _ in range(2):
    tr.d_step(real, mixing_noise(4, 512, cfg.mixing, 'cuda'))
    fns[which]()
torch.cuda.synchronize()
counts = collections.Counter()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'rick_amd' in fr.filename:
            return f'{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:70]}'
    return '?'


def wrap(obj, name, pred=lambda *a, **k: True):
    orig = getattr(obj, name)

    def f(*a, **k):
        if pred(*a, **k):
            t = a[0] if a and torch.is_tensor(a[0]) else None
            counts[(name, tuple(t.shape) if t is not None else None, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)


wrap(torch.Tensor, 'clone')
wrap(torch.Tensor, 'copy_')
wrap(torch.Tensor, 'contiguous', lambda t, *a, **k: t.is_cuda and not t.is_contiguous(memory_format=k.get('memory_format', torch.contiguous_format)))
wrap(torch.Tensor, 'zero_')
wrap(torch.Tensor, 'fill_')
wrap(torch.Tensor, 'repeat')
for fn in ('zeros', 'zeros_like', 'ones', 'full', 'cat', 'stack'):
    wrap(torch, fn)
fns[which]()
torch.cuda.synchronize()
for (name, shp, st), n in counts.most_common(40):
    print(f'{n:3d}x {name:11s} {str(shp):24s} {st}')
