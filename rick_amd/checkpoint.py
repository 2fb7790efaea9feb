"""Checkpoint compatibility with the reference script.

The reference stores ``{"g_ema", "g", "d", "g_optim", "d_optim"}`` with plain ``state_dict()``s
(train_dynamic_update_prune.py:647-659) and starts from a rosinality-style source checkpoint through
``load_state_dict(ckpt[...], strict=False)`` for G, g_ema, D and d_ema <- D (:871-879).  Module state
dicts already share the reference's key names (SURVEY §8a row M*); the optimiser state is the one
thing laid out differently here (one flat m / v buffer per network and per-parameter step counts inside
MaskedFlatAdam), so this module converts it to and from ``torch.optim.Adam.state_dict()`` layout:
a checkpoint written here resumes under the reference script and vice versa.
"""
import torch


def adam_state_dict(optim):
    """MaskedFlatAdam -> the dict torch.optim.Adam(params_in_optimiser_order, lr, betas, eps).state_dict() holds
    after the same steps.  Parameters that never stepped (torch: grad was None every time) have no state entry."""
    fp = optim.fp
    state = {}
    for j, i in enumerate(fp.opt_idx):
        if optim.steps[i] == 0:
            continue
        lo, hi = fp.segment(fp.names[i])
        shape = fp.params[i].shape
        state[j] = {'step': torch.tensor(float(optim.steps[i])),
                    'exp_avg': optim.m[lo:hi].view(shape).detach().clone(),
                    'exp_avg_sq': optim.v[lo:hi].view(shape).detach().clone()}
    group = {'lr': optim.lr, 'betas': tuple(optim.betas), 'eps': optim.eps, 'weight_decay': 0, 'amsgrad': False,
             'maximize': False, 'foreach': None, 'capturable': False, 'params': list(range(len(fp.opt_idx)))}
    return {'state': state, 'param_groups': [group]}


def load_adam_state_dict(optim, sd):
    """Inverse of adam_state_dict; accepts a torch.optim.Adam state_dict over the same parameters in the same order
    (the reference's g_probe_params / d_probe_params lists, :908-931)."""
    fp = optim.fp
    groups = sd['param_groups']
    n = sum(len(g['params']) for g in groups)
    if n != len(fp.opt_idx):
        raise RuntimeError(f'optimizer state has {n} parameters, this optimiser owns {len(fp.opt_idx)}')
    ids = [pid for g in groups for pid in g['params']]
    optim.m.zero_()
    optim.v.zero_()
    for j, i in enumerate(fp.opt_idx):
        st = sd['state'].get(ids[j])
        optim.steps[i] = 0
        if st is None:
            continue
        lo, hi = fp.segment(fp.names[i])
        if st['exp_avg'].numel() != hi - lo:
            raise RuntimeError(f'optimizer state of parameter {fp.names[i]} has the wrong size')
        optim.steps[i] = int(round(float(st['step'])))
        optim.m[lo:hi].copy_(st['exp_avg'].reshape(-1))
        optim.v[lo:hi].copy_(st['exp_avg_sq'].reshape(-1))
    optim.sync_steps_to_device()
    g0 = groups[0]
    optim.lr, optim.betas, optim.eps = g0['lr'], tuple(g0['betas']), g0['eps']


def state_dict(trainer):
    """The reference's checkpoint dict (:647-659) for a RickTrainer."""
    getattr(trainer, '_finish_pending', lambda: None)()      # a deferred optimiser step (data-parallel pipelining) lands first
    return {'g_ema': trainer.g_ema.state_dict(), 'g': trainer.g.state_dict(), 'd': trainer.d.state_dict(),
            'g_optim': adam_state_dict(trainer.g_optim), 'd_optim': adam_state_dict(trainer.d_optim)}


def save(trainer, path):
    torch.save(state_dict(trainer), path)


def load_source(ckpt, generator, g_ema, discriminator, d_ema):
    """Start-up load of the reference (:871-879): a source-domain checkpoint dict (or a path to one),
    ``strict=False`` everywhere, d_ema initialised from "d"."""
    if not isinstance(ckpt, dict):
        ckpt = torch.load(ckpt, map_location='cpu')
    generator.load_state_dict(ckpt['g'], strict=False)
    g_ema.load_state_dict(ckpt['g_ema'], strict=False)
    discriminator.load_state_dict(ckpt['d'], strict=False)
    d_ema.load_state_dict(ckpt['d'], strict=False)
    from . import op
    op.bump_weights_epoch()
    return ckpt


def resume(trainer, ckpt):
    """Continue a run from a checkpoint written by `save` or by the reference script (needs "g_optim"/"d_optim")."""
    if not isinstance(ckpt, dict):
        ckpt = torch.load(ckpt, map_location='cpu')
    # a deferred optimiser step (data-parallel pipelining) belongs to the OLD state: it lands before anything is loaded —
    # replayed afterwards it would apply pre-load gradients to the restored weights / moments and bump their step counts
    getattr(trainer, '_finish_pending', lambda: None)()
    load_source(ckpt, trainer.g, trainer.g_ema, trainer.d, trainer.d_ema)
    if 'g_optim' in ckpt:
        load_adam_state_dict(trainer.g_optim, ckpt['g_optim'])
    if 'd_optim' in ckpt:
        load_adam_state_dict(trainer.d_optim, ckpt['d_optim'])
    trainer.invalidate_graphs()      # captured steps bake in the Adam run grouping of the old step counts
    return ckpt
