"""Find the first step of a long graph-mode run that produces a non-finite loss / parameter (GPU).
usage: python tools/debug_nan.py [iterations]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.models import Discriminator, Generator
from rick_amd.synth import synth_reals
from rick_amd.train import RickTrainer, TrainConfig
torch.manual_seed(1)
dev = 'cuda'
cfg = TrainConfig(batch=4)
g, d = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
ge, de = Generator(256, 512, 8).to(dev), Discriminator(256).to(dev)
tr = RickTrainer(cfg, g, d, ge, de)
real = [synth_reals(4, 256, seed=s).to(dev) for s in range(4)]
tr.enable_graphs('--eager' not in sys.argv)
tr.prepare_graphs(real[0])
i0 = cfg.warmup_iter + 1
n = int(([a for a in sys.argv[1:] if a.isdigit()] or ['500'])[0])


def finite(t):
    return bool(torch.isfinite(t).all())


def check(tag, i):
    bad = []
    for name, flat in (('g', tr.g_flat), ('d', tr.d_flat)):
        if hasattr(tr, '_finish_pending'):
            pass
        if not finite(flat.flat):
            bad.append(f'{name}.params')
        if not finite(flat.grad):
            bad.append(f'{name}.grads')
    for k, v in tr.losses.items():
        if torch.is_tensor(v) and not finite(v):
            bad.append(f'loss[{k}]')
    if bad:
        print(f'iteration {i} after {tag}: NON-FINITE {bad}', flush=True)
        # which parameters / gradients
        for name, net, flat in (('g', g, tr.g_flat), ('d', d, tr.d_flat)):
            for pn, p in net.named_parameters():
                if not finite(p.data):
                    print(f'   {name}.{pn}: data', tuple(p.shape))
                if p.grad is not None and not finite(p.grad):
                    nb = int((~torch.isfinite(p.grad)).sum())
                    print(f'   {name}.{pn}: grad {nb} of {p.grad.numel()} non-finite, max finite {float(p.grad[torch.isfinite(p.grad)].abs().max()) if nb < p.grad.numel() else float("nan"):.3e}')
        sys.exit(1)


for k in range(n):
    i = i0 + k
    r = real[k % 4]
    if tr.use_graphs:
        if tr._real is None:
            tr._real = torch.empty_like(r)
        tr._real.copy_(r)
        r = tr._real
    gr = tr.use_graphs
    from rick_amd.train import mixing_noise
    nz = (lambda b: None) if gr else (lambda b: mixing_noise(b, cfg.latent, cfg.mixing, dev))
    tr.d_step(r, nz(4), i, graph=gr); tr._finish_pending() if hasattr(tr, '_finish_pending') else None; check('d', i)
    if i % cfg.d_reg_every == 0:
        tr.r1_step(r, i, graph=gr); tr._finish_pending(); check('r1', i)
    tr.g_step(nz(4), graph=gr); tr._finish_pending(); check('g', i)
    if i % cfg.g_reg_every == 0:
        tr.plr_step(nz(2), graph=gr); tr._finish_pending(); check('plr', i)
    tr.ema_step()
    if k % 50 == 0:
        print(k, {kk: round(float(v), 4) for kk, v in tr.losses.items() if torch.is_tensor(v) and v.numel() == 1}, 'max|w| g', float(tr.g_flat.flat.abs().max()), 'd', float(tr.d_flat.flat.abs().max()), flush=True)
print('finite through', n)
