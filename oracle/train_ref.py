"""Oracle (test infrastructure): losses, gradient penalties, Fisher estimate and the
freeze / fine-tune / prune decisions of train_dynamic_update_prune.py, restated
for CPU (PyTorch + NumPy).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .model_ref import discriminator_ref, generator_ref


def d_logistic_loss_ref(real_pred, fake_pred):
    """train_dynamic_update_prune.py:82-86"""
    return F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()


def g_nonsaturating_loss_ref(fake_pred):
    """train_dynamic_update_prune.py:99-101"""
    return F.softplus(-fake_pred).mean()


def d_r1_loss_ref(real_pred, real_img):
    """train_dynamic_update_prune.py:89-96"""
    (g,) = torch.autograd.grad(real_pred.sum(), real_img, create_graph=True)
    return g.pow(2).reshape(g.shape[0], -1).sum(1).mean()


def g_path_regularize_ref(fake_img, latents, mean_path_length, noise, decay=0.01):
    """train_dynamic_update_prune.py:104-118.  `noise` is the randn_like(fake_img)
    draw (passed in so both sides of a parity test share it)."""
    noise = noise / math.sqrt(fake_img.shape[2] * fake_img.shape[3])
    (g,) = torch.autograd.grad((fake_img * noise).sum(), latents, create_graph=True)
    path_lengths = torch.sqrt(g.pow(2).sum(2).mean(1))
    path_mean = mean_path_length + decay * (path_lengths.mean() - mean_path_length)
    penalty = (path_lengths - path_mean).pow(2).mean()
    return penalty, path_mean.detach(), path_lengths


def fisher_sample_ref(sd_g, sd_d, z, real, size=256):
    """One Fisher sample (train_dynamic_update_prune.py:231-248 with
    model_probe_tune.py:481-504, :706-729): grad^2 of the G / D losses w.r.t.
    every parameter, batch 1, fixed noise buffers unless randomised by caller."""
    pg = {k: v.detach().clone().requires_grad_(True) for k, v in sd_g.items() if _is_param(k)}
    pd = {k: v.detach().clone().requires_grad_(True) for k, v in sd_d.items() if _is_param(k)}
    full_g = dict(sd_g); full_g.update(pg)
    full_d = dict(sd_d); full_d.update(pd)
    fake, _ = generator_ref(full_g, [z.reshape(1, -1)], size=size, randomize_noise=False)
    fake_pred, _ = discriminator_ref(full_d, fake, size=size)
    real_pred, _ = discriminator_ref(full_d, real.reshape(1, 3, size, size), size=size)
    g_loss = g_nonsaturating_loss_ref(fake_pred)
    d_loss = d_logistic_loss_ref(real_pred, fake_pred)
    gg = torch.autograd.grad(g_loss, list(pg.values()), retain_graph=True, allow_unused=True)
    gd = torch.autograd.grad(d_loss, list(pd.values()), retain_graph=True, allow_unused=True)
    fg = {k: (g.detach() ** 2 if g is not None else torch.zeros_like(p))
          for (k, p), g in zip(pg.items(), gg)}
    fd = {k: (g.detach() ** 2 if g is not None else torch.zeros_like(p))
          for (k, p), g in zip(pd.items(), gd)}
    return fg, fd, float(g_loss), float(d_loss)


def _is_param(key):
    return not (key.endswith('.kernel') or key.startswith('noises.'))


# ------------------------------------------------------------------ per-filter FIM + decisions

def g_filter_fim_ref(fisher_g, n_blocks=12):
    """Per-filter Fisher of the generator (train_dynamic_update_prune.py:279-299):
    conv:  mean over (0,2,3,4) of convs.k.conv.weight  -> [Co]
    fc:    (modulation.weight.mean(1) + modulation.bias)/2 -> [Ci]"""
    conv, fc = {}, {}
    for k in range(n_blocks):
        conv[f'convs.{k}.conv.weight'] = np.asarray(fisher_g[f'convs.{k}.conv.weight']).mean(axis=(0, 2, 3, 4))
        w = np.asarray(fisher_g[f'convs.{k}.conv.modulation.weight']).mean(axis=1)
        b = np.asarray(fisher_g[f'convs.{k}.conv.modulation.bias'])
        fc[f'convs.{k}.conv.modulation.weight'] = (w + b) / 2
    return conv, fc


def d_filter_fim_ref(fisher_d, blocks=range(1, 7)):
    """Per-filter Fisher of the discriminator (train_dynamic_update_prune.py:334-353)."""
    out = {}
    for b in blocks:
        for li in range(2):
            wk = f'convs.{b}.conv{li + 1}.{li}.weight'
            bk = f'convs.{b}.conv{li + 1}.{li + 1}.bias'
            out[wk] = (np.asarray(fisher_d[wk]).mean(axis=(1, 2, 3)) + np.asarray(fisher_d[bk])) / 2
            if li == 1:
                sk = f'convs.{b}.skip.{li}.weight'
                out[sk] = np.asarray(fisher_d[sk]).mean(axis=(1, 2, 3))
    return out


def _split(fim, cut, prune, skip_rule=False):
    if skip_rule:   # train_dynamic_update_prune.py:382-384 (>= / < on the prune line)
        return (np.where(fim > cut)[0], np.where((fim >= prune) & (fim <= cut))[0], np.where(fim < prune)[0])
    return (np.where(fim > cut)[0], np.where((fim > prune) & (fim <= cut))[0], np.where(fim <= prune)[0])


def g_decisions_ref(fisher_g, fisher_quantile, prune_quantile, n_blocks=12):
    """train_dynamic_update_prune.py:279-330.  Returns (freeze, ft, prune) dicts of index arrays."""
    conv, fc = g_filter_fim_ref(fisher_g, n_blocks)
    allc = np.concatenate([[]] + [conv[k] for k in conv], axis=None)
    allf = np.concatenate([[]] + [fc[k] for k in fc], axis=None)
    cut_c, pr_c = np.percentile(allc, q=fisher_quantile), np.percentile(allc, q=prune_quantile)
    cut_f, pr_f = np.percentile(allf, q=fisher_quantile), np.percentile(allf, q=prune_quantile)
    freeze, ft, prune = {}, {}, {}
    for k, v in conv.items():
        freeze[k], ft[k], prune[k] = _split(v, cut_c, pr_c)
    for k, v in fc.items():
        for kk in (k, k.replace('weight', 'bias')):
            freeze[kk], ft[kk], prune[kk] = _split(v, cut_f, pr_f)
    return freeze, ft, prune


def d_decisions_ref(fisher_d, fisher_quantile, prune_quantile, blocks=range(1, 7)):
    """train_dynamic_update_prune.py:334-384."""
    fim = d_filter_fim_ref(fisher_d, blocks)
    allv = np.concatenate([[]] + [fim[k] for k in fim], axis=None)
    cut, pr = np.percentile(allv, q=fisher_quantile), np.percentile(allv, q=prune_quantile)
    freeze, ft, prune = {}, {}, {}
    for k, v in fim.items():
        if 'skip' in k:
            freeze[k], ft[k], prune[k] = _split(v, cut, pr, skip_rule=True)
        else:
            bk = k.replace(f'{k[-8]}.weight', f'{int(k[-8]) + 1}.bias')
            for kk in (k, bk):
                freeze[kk], ft[kk], prune[kk] = _split(v, cut, pr)
    return freeze, ft, prune


def zero_idx_merge_ref(old, new):
    """train_dynamic_update_prune.py:138-144"""
    return {k: np.unique(np.concatenate((old[k], new[k]))) for k in old}


def adam_step_ref(p, g, m, v, step, lr, beta1, beta2, eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad) single-tensor math, as configured at
    train_dynamic_update_prune.py:913-931."""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v
