"""Host side of the RICK adaptation loop on the MI355X engine: losses and gradient penalties,
the Fisher-information sweep with on-device grad^2 accumulation and per-filter reduction, the
freeze / fine-tune / prune decisions and masks, the masked flat Adam + EMA, and the per-iteration
control flow of ``train()`` (train_dynamic_update_prune.py:159-699).

Work the reference performs and then discards is skipped (SURVEY.md §8a footnote): the D step
runs on ``fake.detach()``, the G step does not compute D weight gradients, and parameters that
no optimiser owns do not get gradients.  Results are unchanged.
"""
import math
import os
import random
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F
from torch import autograd

from . import op
from ._lib import check, lib, ptr, stream_ptr


# --------------------------------------------------------------------------- losses
def d_logistic_loss(real_pred, fake_pred):
    """train_dynamic_update_prune.py:82-86"""
    return F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()


def g_nonsaturating_loss(fake_pred):
    """train_dynamic_update_prune.py:99-101"""
    return F.softplus(-fake_pred).mean()


def d_r1_loss(real_pred, real_img):
    """train_dynamic_update_prune.py:89-96 (needs op.second_order())."""
    with op.no_param_grads():      # the inner gradient is w.r.t. the image only: no weight-gradient launches for it
        (grad_real,) = autograd.grad(outputs=real_pred.sum(), inputs=real_img, create_graph=True)
    return grad_real.pow(2).reshape(grad_real.shape[0], -1).sum(1).mean()


def g_path_regularize(fake_img, latents, mean_path_length, decay=0.01, noise=None):
    """train_dynamic_update_prune.py:104-118 (needs op.second_order()).  `noise` overrides the
    randn_like draw so parity tests can share it with the oracle."""
    if noise is None:
        noise = torch.randn_like(fake_img)
    noise = noise / math.sqrt(fake_img.shape[2] * fake_img.shape[3])
    with op.no_param_grads():      # the inner gradient is w.r.t. the latents only
        (grad,) = autograd.grad(outputs=(fake_img * noise).sum(), inputs=latents, create_graph=True)
    path_lengths = torch.sqrt(grad.pow(2).sum(2).mean(1))
    path_mean = mean_path_length + decay * (path_lengths.mean() - mean_path_length)
    path_penalty = (path_lengths - path_mean).pow(2).mean()
    return path_penalty, path_mean.detach(), path_lengths


def make_noise(batch, latent_dim, n_noise, device):
    if n_noise == 1:
        return torch.randn(batch, latent_dim, device=device)
    return torch.randn(n_noise, batch, latent_dim, device=device).unbind(0)


def mixing_noise(batch, latent_dim, prob, device):
    """train_dynamic_update_prune.py:130-135"""
    if prob > 0 and random.random() < prob:
        return make_noise(batch, latent_dim, 2, device)
    return [make_noise(batch, latent_dim, 1, device)]


def requires_grad(model, flag=True, only=None):
    for name, p in model.named_parameters():
        if only is None or only(name):
            p.requires_grad = flag


# ------------------------------------------------------------------ flat params / Adam / EMA
FLAT_ALIGN = 64      # floats


class FlatParams:
    """Re-homes the parameters of a network into ONE flat fp32 buffer (each parameter becomes a view)
    with a matching flat gradient buffer (``p.grad`` are views too).  `opt_filter` marks the parameters an
    optimiser owns; they must be contiguous in registration order (true for both reference networks:
    ``convs.*`` of G, ``convs.1-6 + final_*`` of D) so that one kernel covers mask + Adam of the whole
    slice, one memset zeroes its gradients, RCCL reduces contiguous buckets, and one kernel does the EMA
    of the entire network."""

    def __init__(self, named_params, opt_filter=None):
        named_params = list(named_params)
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        sizes = [p.numel() for p in self.params]
        # every parameter starts on a 256-byte boundary (a bias that follows the 1-element noise strength would
        # otherwise be misaligned for the float4 kernels); offsets[i + 1] is the START of parameter i + 1, the
        # padding in between stays zero (zero gradient: Adam, EMA and the all-reduce leave it zero)
        self.sizes = sizes
        starts, pos = [], 0
        for n in sizes:
            starts.append(pos)
            pos += (n + FLAT_ALIGN - 1) // FLAT_ALIGN * FLAT_ALIGN
        self.offsets = np.array(starts + [pos], dtype=np.int64)
        self.total = int(self.offsets[-1])
        dev = self.params[0].device
        self.flat = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(self.total, device=dev, dtype=torch.float32)
        for p, o, n in zip(self.params, self.offsets, sizes):
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
        self.index = {n: i for i, n in enumerate(self.names)}
        self.opt_idx = [i for i, n in enumerate(self.names) if opt_filter is None or opt_filter(n)]
        if self.opt_idx and self.opt_idx != list(range(self.opt_idx[0], self.opt_idx[-1] + 1)):
            raise RuntimeError('FlatParams: optimised parameters must be contiguous in registration order')
        self.lo = int(self.offsets[self.opt_idx[0]]) if self.opt_idx else 0
        self.hi = int(self.offsets[self.opt_idx[-1] + 1]) if self.opt_idx else 0

    def zero_grad(self):
        self.grad[self.lo:self.hi].zero_()

    def segment(self, name):
        i = self.index[name]
        return int(self.offsets[i]), int(self.offsets[i]) + self.sizes[i]


class MaskedFlatAdam:
    """torch.optim.Adam(lr, betas, eps=1e-8) over the optimised slice of a FlatParams, with RICK's
    freeze / prune masks applied in the same kernel (train_dynamic_update_prune.py:427-438, 522-540).
    Parameters whose ``requires_grad`` is False at step time are skipped exactly like torch skips
    ``grad is None`` (per-parameter step counts, used by the warm-up stage :202-211)."""

    def __init__(self, flat, lr, betas, eps=1e-8):
        self.fp, self.lr, self.betas, self.eps = flat, lr, betas, eps
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.steps = [0] * len(flat.params)          # host mirror of steps_dev (checkpoints, run grouping)
        # step counters and bias corrections also live on the device (rick_adam_prepare_f32), so an optimiser step
        # depends on device state only and a captured hipGraph of a whole train step can be replayed
        self.steps_dev = torch.zeros(len(flat.params), device=flat.flat.device, dtype=torch.int32)
        self.bc = torch.ones(8, 2, device=flat.flat.device, dtype=torch.float32)      # one row per run of equal step count
        self.mask = None            # uint8 flat: bit0 freeze, bit1 prune
        self.last_runs = []         # [(first param, last param)] of the most recent step()

    def set_mask(self, mask):
        if self.mask is not None and self.mask.shape == mask.shape:
            self.mask.copy_(mask)   # in place: launches captured in a graph keep reading the same buffer
        else:
            self.mask = mask

    def sync_steps_to_device(self):
        self.steps_dev.copy_(torch.tensor(self.steps, dtype=torch.int32))

    def note_replayed_step(self):
        """A captured step() was replayed: the device counters advanced, mirror it on the host."""
        for i0, i1 in self.last_runs:
            for i in range(i0, i1 + 1):
                self.steps[i] += 1

    def step(self):
        fp = self.fp
        idx = fp.opt_idx
        active = {i: fp.params[i].requires_grad for i in idx}
        k, n = 0, len(idx)
        runs = []
        while k < n:
            i = idx[k]
            if not active[i]:
                k += 1
                continue
            kk = k
            self.steps[i] += 1
            while kk + 1 < n and active[idx[kk + 1]] and self.steps[idx[kk + 1]] + 1 == self.steps[i]:
                kk += 1
                self.steps[idx[kk]] += 1
            lo, hi = int(fp.offsets[i]), int(fp.offsets[idx[kk] + 1])
            b1, b2 = self.betas
            mk = None if self.mask is None else self.mask[lo:hi]
            if len(runs) >= self.bc.shape[0]:
                raise RuntimeError('MaskedFlatAdam: too many runs of distinct step counts')
            bc = self.bc[len(runs)]
            check(lib.rick_adam_prepare_f32(ptr(self.steps_dev), i, idx[kk] - i + 1, b1, b2, ptr(bc), stream_ptr()),
                  'rick_adam_prepare_f32')
            from .op.conv import hbm_launch
            check(hbm_launch('masked_adam', (28 + (1 if mk is not None else 0)) * (hi - lo), lib.rick_masked_adam_dev_f32,
                             ptr(fp.flat[lo:hi]), ptr(fp.grad[lo:hi]), ptr(self.m[lo:hi]), ptr(self.v[lo:hi]), ptr(mk), hi - lo,
                             self.lr, b1, b2, self.eps, ptr(bc), stream_ptr()), 'rick_masked_adam_dev_f32')
            runs.append((i, idx[kk]))
            k = kk + 1
        self.last_runs = runs
        op.bump_weights_epoch(fp.params)


def ema_flat(ema_flat_params, flat_params, decay):
    """accumulate() of the reference (train_dynamic_update_prune.py:68-73) over every parameter of a
    network in ONE launch (both sides are FlatParams with the same layout)."""
    if ema_flat_params.total != flat_params.total:
        raise RuntimeError('ema_flat: layouts differ')
    check(lib.rick_ema_f32(ptr(ema_flat_params.flat), ptr(flat_params.flat), flat_params.total, decay, stream_ptr()),
          'rick_ema_f32')
    op.bump_weights_epoch(ema_flat_params.params)


# ------------------------------------------------------------------------ Fisher sweep
class FisherAccumulator:
    """Device-side replacement for the per-sample ``.cpu().numpy()`` accumulation of grad^2
    (train_dynamic_update_prune.py:252-263): acc += g^2 in one fused kernel per tensor."""

    def __init__(self, named_params):
        self.names = [n for n, _ in named_params]
        self.acc = {n: torch.zeros_like(p, memory_format=torch.contiguous_format) for n, p in named_params}

    def add(self, grads):
        for n, g in zip(self.names, grads):
            if g is None:
                continue
            g = g.contiguous()
            check(lib.rick_sq_accumulate_f32(ptr(self.acc[n]), ptr(g), g.numel(), stream_ptr()),
                  'rick_sq_accumulate_f32')

    def scale_(self, s):
        for v in self.acc.values():
            v.mul_(s)

    def zero_(self):
        for v in self.acc.values():
            v.zero_()


def filter_mean(t, filter_dim):
    """Mean over every dim except `filter_dim` with one wavefront per filter
    (rick_filter_reduce_f32).  Supports filter_dim 0 ([F, ...]) and 1 of a [1, F, ...] tensor."""
    t = t.contiguous()
    if filter_dim == 1:
        if t.shape[0] != 1:
            raise RuntimeError('filter_mean: dim-1 filters need a leading dim of 1')
        t = t[0]
    nf = t.shape[0]
    inner = t.numel() // nf
    out = torch.empty(nf, device=t.device, dtype=t.dtype)
    check(lib.rick_filter_reduce_f32(ptr(t), ptr(out), 1, 0, nf, inner, inner, 1.0 / inner, stream_ptr()),
          'rick_filter_reduce_f32')
    return out


def g_filter_fim(fisher, n_blocks=None):
    """Per-filter FIM of the generator (train_dynamic_update_prune.py:279-299): dicts of device vectors."""
    conv, fc = {}, {}
    if n_blocks is None:      # 12 at 256 px, the only size the reference supports (hard-coded range(12), :281)
        n_blocks = sum(1 for k in fisher if k.startswith('convs.') and k.endswith('.conv.weight'))
    for k in range(n_blocks):
        conv[f'convs.{k}.conv.weight'] = filter_mean(fisher[f'convs.{k}.conv.weight'], 1)
        wk = f'convs.{k}.conv.modulation.weight'
        fc[wk] = (filter_mean(fisher[wk], 0) + fisher[f'convs.{k}.conv.modulation.bias']) / 2
    return conv, fc


def d_filter_fim(fisher, blocks=None):
    """Per-filter FIM of the discriminator (train_dynamic_update_prune.py:334-353)."""
    out = {}
    if blocks is None:        # range(1, 7) at 256 px (:336)
        blocks = range(1, 1 + sum(1 for k in fisher if k.endswith('.skip.1.weight')))
    for b in blocks:
        for li in range(2):
            wk, bk = f'convs.{b}.conv{li + 1}.{li}.weight', f'convs.{b}.conv{li + 1}.{li + 1}.bias'
            out[wk] = (filter_mean(fisher[wk], 0) + fisher[bk]) / 2
            if li == 1:
                sk = f'convs.{b}.skip.{li}.weight'
                out[sk] = filter_mean(fisher[sk], 0)
    return out


def _split(fim, cut, prune, skip_rule=False):
    if skip_rule:   # train_dynamic_update_prune.py:382-384
        return np.where(fim > cut)[0], np.where((fim >= prune) & (fim <= cut))[0], np.where(fim < prune)[0]
    return np.where(fim > cut)[0], np.where((fim > prune) & (fim <= cut))[0], np.where(fim <= prune)[0]


def decide_g(conv, fc, fisher_quantile, prune_quantile):
    """freeze / ft / prune index sets of the generator from per-filter FIM vectors (host numpy,
    a few thousand floats; np.percentile in PERCENT units — train_dynamic_update_prune.py:285-330)."""
    conv = {k: np.asarray(v, dtype=np.float32) for k, v in conv.items()}
    fc = {k: np.asarray(v, dtype=np.float32) for k, v in fc.items()}
    allc = np.concatenate([[]] + list(conv.values()), axis=None)
    allf = np.concatenate([[]] + list(fc.values()), axis=None)
    cut_c, pr_c = np.percentile(allc, q=fisher_quantile), np.percentile(allc, q=prune_quantile)
    cut_f, pr_f = np.percentile(allf, q=fisher_quantile), np.percentile(allf, q=prune_quantile)
    freeze, ft, prune = {}, {}, {}
    for k, v in conv.items():
        freeze[k], ft[k], prune[k] = _split(v, cut_c, pr_c)
    for k, v in fc.items():
        for kk in (k, k.replace('weight', 'bias')):
            freeze[kk], ft[kk], prune[kk] = _split(v, cut_f, pr_f)
    return freeze, ft, prune


def decide_d(fim, fisher_quantile, prune_quantile):
    """train_dynamic_update_prune.py:352-384"""
    fim = {k: np.asarray(v, dtype=np.float32) for k, v in fim.items()}
    allv = np.concatenate([[]] + list(fim.values()), axis=None)
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


def zero_idx_merge(old, new):
    """train_dynamic_update_prune.py:138-144"""
    return {k: np.unique(np.concatenate((old[k], new[k]))) for k in old}


def build_mask(flat, freeze_idx, zero_idx):
    """Flat uint8 mask for MaskedFlatAdam from per-key filter index sets: bit0 = freeze
    (grad := 0), bit1 = zero/prune (param := 0, grad := 0).  5-D generator conv weights are
    indexed on dim 1, everything else on dim 0 (train_dynamic_update_prune.py:524-537, 429-435)."""
    mask = np.zeros(flat.total, dtype=np.uint8)      # plain NumPy on the host: a few thousand contiguous filter rows
    for bit, table in ((1, freeze_idx), (2, zero_idx)):
        for name, idx in table.items():
            if name not in flat.index or len(idx) == 0:
                continue
            lo, hi = flat.segment(name)
            p = flat.params[flat.index[name]]
            view = mask[lo:hi].reshape(tuple(p.shape))
            ii = np.asarray(idx, dtype=np.int64)
            if p.ndim == 5:
                view[:, ii] |= bit
            else:
                view[ii] |= bit
    return torch.from_numpy(mask).to(flat.flat.device)


# --------------------------------------------------------------------------- trainer
@dataclass
class TrainConfig:
    """The live flags of the reference CLI (train_dynamic_update_prune.py:703-758) with its defaults."""
    size: int = 256
    batch: int = 4
    latent: int = 512
    n_mlp: int = 8
    channel_multiplier: int = 2
    r1: float = 10.0
    path_regularize: float = 2.0
    path_batch_shrink: int = 2
    d_reg_every: int = 16
    g_reg_every: int = 4
    mixing: float = 0.9
    lr: float = 0.002
    num_fisher_img: int = 5
    fisher_freq: int = 50
    fisher_quantile: float = 40.0
    prune_quantile: float = 0.1
    warmup_iter: int = 250
    ema_decay: float = 0.5 ** (32 / (10 * 1000))     # train_dynamic_update_prune.py:180


def g_optim_filter(name):
    return 'convs' in name                              # train_dynamic_update_prune.py:909-911


def d_optim_filter(name):
    return ('convs' in name and 'convs.0' not in name) or 'final' in name   # :922-926


class RickTrainer:
    """One process / one GPU worth of the adaptation loop.  `dp` (rick_amd.dist.DataParallelGrads
    or None) averages flat gradients over ranks with RCCL before each optimiser step."""

    def __init__(self, cfg, generator, discriminator, g_ema, d_ema, dp=None):
        self.cfg, self.g, self.d, self.g_ema, self.d_ema, self.dp = cfg, generator, discriminator, g_ema, d_ema, dp
        self.device = next(generator.parameters()).device
        # packed conv weights: one refresh launch per network
        self._pack_groups = [op.register_pack_group(net) for net in (generator, discriminator)]
        self._ema_pack_groups = [op.register_pack_group(net) for net in (g_ema, d_ema)]
        self.g_flat = FlatParams(generator.named_parameters(), g_optim_filter)
        self.d_flat = FlatParams(discriminator.named_parameters(), d_optim_filter)
        self.g_ema_flat = FlatParams(g_ema.named_parameters())
        self.d_ema_flat = FlatParams(d_ema.named_parameters())
        g_ratio = cfg.g_reg_every / (cfg.g_reg_every + 1)
        d_ratio = cfg.d_reg_every / (cfg.d_reg_every + 1)
        self.g_optim = MaskedFlatAdam(self.g_flat, cfg.lr * g_ratio, (0 ** g_ratio, 0.99 ** g_ratio))
        self.d_optim = MaskedFlatAdam(self.d_flat, cfg.lr * d_ratio, (0 ** d_ratio, 0.99 ** d_ratio))
        # parameters no optimiser owns never need gradients in the train steps
        for n, p in generator.named_parameters():
            p.requires_grad = g_optim_filter(n)
        self._d_all = dict(discriminator.named_parameters())
        self.mean_path_length = 0
        self.idx_freeze_g = self.idx_freeze_d = None
        self.zero_idx_g = self.zero_idx_d = None
        self.losses = {}
        self.use_graphs = False
        self.step_events = None     # list of (step name, event) while bench.py times step types
        self._gs, self._inject, self._layer_idx, self._real = {}, {}, None, None
        self._fisher_state = None   # persistent accumulators / static inputs / captured per-sample graph of the Fisher sweep
        self._pending = None        # (graph state, flat, optimiser, optimiser graph) of a step whose gradient exchange is in flight
        if dp is not None:
            dp.attach(self.g_flat, self.d_flat)

    # ---- gradient flags of the discriminator for the current stage (:202-211)
    def _set_d_stage(self, i):
        warm = i < self.cfg.warmup_iter
        for n, p in self._d_all.items():
            p.requires_grad = (('final' in n) if warm else True) and d_optim_filter(n)

    def _d_frozen(self):
        class _Ctx:
            def __init__(s, params):
                s.params, s.flags = params, None

            def __enter__(s):
                s.flags = [p.requires_grad for p in s.params]
                for p in s.params:
                    p.requires_grad = False

            def __exit__(s, *a):
                for p, f in zip(s.params, s.flags):
                    p.requires_grad = f
        return _Ctx(list(self._d_all.values()))

    def _reduce(self, flat):
        if self.dp is not None:
            self.dp.all_reduce(flat)

    def _sink(self):
        """op.grad_sink() — except in the eager data-parallel mode, whose bucket all-reduces are launched from the
        parameters' post-accumulate-grad hooks while backward is still running: a sunk gradient fires no hook, so every
        bucket with a conv weight would only leave at the end of backward and nothing would overlap."""
        import contextlib
        if self.dp is not None and getattr(self.dp, 'hooks_enabled', False) and getattr(self.dp, 'active', getattr(self.dp, 'world', 1) > 1):
            return contextlib.nullcontext()
        return op.grad_sink()

    def _zero_grad(self, flat):
        flat.zero_grad()
        if self.dp is not None:
            self.dp.prepare(flat)          # bucket counts follow the stage's requires_grad flags

    # ---- hipGraph replay of whole steps ------------------------------------------------------------------
    # One RICK iteration is ~2 500 kernel launches issued from Python autograd; measured on MI355X the host needs
    # longer to ENQUEUE an iteration (34.8 ms) than the GPU needs to run it, so the loop was host-bound.  With
    # `use_graphs` every step type (D, R1, G, path length) is captured once — forward, backward, weight re-packing
    # and the masked Adam with its device-side step counters — and replayed afterwards; per-step host randomness
    # (style-mixing index) is written into device scalars before the replay.  With data parallelism the capture is
    # split around the gradient all-reduce (forward/backward graph -> RCCL -> optimiser graph).
    def enable_graphs(self, on=True):
        self.use_graphs = bool(on)
        if self.dp is not None:
            self.dp.hooks_enabled = not self.use_graphs      # a replayed backward runs no Python hooks
        if on:
            for opt in (self.g_optim, self.d_optim):         # the mask buffer must exist before a launch is captured
                if opt.mask is None:
                    opt.mask = torch.zeros(opt.fp.total, dtype=torch.uint8, device=self.device)

    def prepare_graphs(self, real_img):
        """Capture all four step graphs up front (two eager warm-up calls + the capturing call of each step type), so
        that no capture lands inside a timed region.  These are real training steps."""
        if not self.use_graphs:
            return
        if self._real is None:
            self._real = torch.empty_like(real_img)
        self._real.copy_(real_img)
        for _ in range(3):
            self.d_step(self._real, None, graph=True)
            self.r1_step(self._real, graph=True)
            self.g_step(None, graph=True)
            self.plr_step(None, graph=True)
            self.ema_step()

    def _draw_inject(self, key):
        """Host side of mixing_noise + Generator.forward's random inject_index (:130-135, model_probe_tune.py:555-560):
        with probability `mixing` two latents, switched at a uniform layer index in [1, n_latent-1]; otherwise one
        latent for every layer (inject = n_latent).  Written to a device scalar the captured graph reads."""
        n_latent = self.g.n_latent
        k = random.randint(1, n_latent - 1) if (self.cfg.mixing > 0 and random.random() < self.cfg.mixing) else n_latent
        t = self._inject.get(key)
        if t is None:
            t = self._inject[key] = torch.zeros((), dtype=torch.int64, device=self.device)
            self._layer_idx = torch.arange(n_latent, device=self.device).view(1, -1, 1)
        t.fill_(k)
        self._advance_latent_pool(key)

    # The mapping network is frozen (the optimiser owns none of its parameters, train_dynamic_update_prune.py:908-917) and its
    # input is fresh noise: the W-space rows of the next LATENT_POOL steps of a step type are i.i.d. draws that do not depend
    # on anything the training changes.  They are computed LATENT_POOL steps at a time by ONE pass through the 8 layers (eagerly,
    # between graph replays) and a step's graph gathers its rows by a device-side index — instead of a randn + 8 mapping-layer
    # launches of ~9-12 us inside every D, G and path-length step (0.2 ms of an iteration).
    LATENT_POOL = 16

    def _pool_ok(self):
        return self.LATENT_POOL > 1 and not os.environ.get('RICK_NO_LATENT_POOL') and not any(
            p.requires_grad for p in self.g.style.parameters())

    def _style_stamp(self):
        """What the pooled rows are valid for: the mapping network's weights as loaded (load_state_dict copies in place and
        bumps _version; the optimiser never touches them)."""
        return tuple((None if p.is_inference() else p._version, p.data_ptr()) for p in self.g.style.parameters())

    def _fill_latent_pool(self, ent):
        with torch.no_grad():
            z = torch.randn(self.LATENT_POOL * ent['rows'], self.cfg.latent, device=self.device)
            ent['w'].copy_(self.g.style(z).view(self.LATENT_POOL, ent['rows'], -1))
        ent['stamp'] = self._style_stamp()

    def _advance_latent_pool(self, key):
        """Host side, before a step's graph runs (with _draw_inject): point the step at the next pool row, refill when used up."""
        ent = getattr(self, '_lat_pool', {}).get(key)
        if ent is None:
            return
        ent['pos'] += 1
        if ent['pos'] >= self.LATENT_POOL or ent.get('stamp') != self._style_stamp():
            # used up — or the mapping network was reloaded (checkpoint.resume / load_source on a trainer that has already
            # stepped): rows computed with the old weights are not draws of the current model
            self._fill_latent_pool(ent)
            ent['pos'] = 0
        ent['idx'].fill_(ent['pos'])

    def _graph_latents(self, key, batch):
        """[batch, n_latent, 512] W-space latents with the style switch applied on the device (same values as the
        reference's cat of repeated w1 / w2 rows)."""
        if not self._pool_ok():
            z = torch.randn(2 * batch, self.cfg.latent, device=self.device)
            w = self.g.style(z).view(2, batch, 1, -1)
            return torch.where(self._layer_idx < self._inject[key], w[0], w[1])
        pools = self.__dict__.setdefault('_lat_pool', {})
        ent = pools.get(key)
        if ent is None or ent['rows'] != 2 * batch:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('latent pool must exist before hipGraph capture (run the step eagerly once)')
            ent = pools[key] = dict(rows=2 * batch, pos=0, idx=torch.zeros(1, dtype=torch.int64, device=self.device),
                                    w=torch.empty(self.LATENT_POOL, 2 * batch, self.cfg.latent, device=self.device))
            self._fill_latent_pool(ent)
        w = ent['w'].index_select(0, ent['idx']).view(2, batch, 1, -1)
        return torch.where(self._layer_idx < self._inject[key], w[0], w[1])

    def _run(self, key, fb, flat, optim, pre=None, fb_head=None):
        """fb(): forward + backward into flat.grad;  then gradient exchange and optimiser step.

        fb_head(): an optional first part of the step that does not read the parameters the PREVIOUS step is still
        updating (the generator forward of the G step does not touch D).  Under data parallelism with step graphs the
        optimiser part of a step is deferred (`_pending`): its gradient buckets are launched right behind the replayed
        forward/backward graph and the step returns; the next step replays its head, THEN waits for the exchange, replays the
        pending optimiser graph and continues — the all-reduce of the D gradients travels while the generator forward
        runs (north star: "overlapped with the next micro-batch forward")."""
        if pre is not None:
            pre()
        st = None
        if key is not None and self.use_graphs:
            st = self._gs.setdefault(key, {'n': 0})
            if 'graphs' in st and st['sig'] != self._graph_signature(optim):
                # optimiser state / stage changed under the capture: drop it AND run the two eager warm-up steps again —
                # a new requires_grad pattern needs new descriptor tables (PackGroup, ModulationBank), which must not be
                # built inside a capture
                self._finish_pending()
                del st['graphs']
                st['n'] = 0
            st['n'] += 1
        if st is None or st['n'] <= 2:                        # eager (also the warm-up of a graph: caches, allocator)
            self._finish_pending()
            if fb_head is not None:
                fb_head()
            fb()
            self._reduce(flat)
            optim.step()
            return
        split = self.dp is not None and getattr(self.dp, 'active', True)
        if 'graphs' not in st:
            self._finish_pending()
            # packed weights are refreshed on the HOST side of the graphs: a network is repacked once per update of its
            # weights (the D step's graph used to repack G again although the G step's graph had just done so, and vice versa)
            for grp in self._pack_groups:
                grp.refresh()
            torch.cuda.synchronize()
            before = list(optim.steps)
            seen = dict(self.losses)
            if not split:
                g = torch.cuda.CUDAGraph()
                with self._capture(g):
                    if fb_head is not None:
                        fb_head()
                    fb()
                    optim.step()
                st['graphs'] = (None, g, None)
            else:
                # head | forward/backward | optimiser as separate graphs from ONE memory pool: the autograd graph the head
                # builds (saved activations in pool memory) is consumed by the backward captured in the second graph
                gh = torch.cuda.CUDAGraph() if fb_head is not None else None
                g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                pool = None
                if gh is not None:
                    with self._capture(gh):
                        fb_head()
                    pool = gh.pool()
                with self._capture(g1, pool=pool):
                    fb()
                with self._capture(g2, pool=g1.pool()):
                    optim.step()
                st['graphs'] = (gh, g1, g2)
            st['runs'] = list(optim.last_runs)
            st['losses'] = {k: v for k, v in self.losses.items() if seen.get(k) is not v}   # this graph's output tensors
            optim.steps[:] = before                           # capture executes nothing: the replay below is this step
            st['sig'] = self._graph_signature(optim)
        gh, g1, g2 = st['graphs']
        if not split:
            for grp in self._pack_groups:
                grp.refresh()
            g1.replay()
            self._after_optimizer(st, flat, optim)
            return
        if gh is not None:
            # The head may only run ahead of a pending step that updates a DIFFERENT network (the generator forward while
            # D's buckets travel).  A pending step on the head's own parameters (two G steps in a row through the step API)
            # must land first: the head would otherwise read pre-update weights that the backward then no longer matches,
            # and its replay would run before the optimiser graph captured behind it in the shared memory pool.
            if self._pending is not None and self._pending[1] is flat:
                self._finish_pending()
            for grp in self._pack_groups:                     # (the head only reads networks no pending step is updating)
                grp.refresh(skip=self._pending_params())
            gh.replay()                                       # ... while the previous step's buckets are on the wire
        self._finish_pending()
        for grp in self._pack_groups:
            grp.refresh()
        g1.replay()
        self.dp.launch(flat)
        self._pending = (st, flat, optim, g2)

    def _capture(self, graph, pool=None):
        """torch.cuda.graph(...) — with a process group alive, in 'thread_local' capture-error mode: RCCL's watchdog thread
        polls its events (hipEventQuery) at any time, and under the default 'global' mode such a call from ANOTHER thread
        while this one is capturing is an error that takes the process down (hipErrorStreamCaptureUnsupported: seen on
        MI355X with one rank; it is a timing matter, so it would have surfaced on some rank of an 8-GPU run)."""
        import torch.distributed as dist
        mode = 'thread_local' if (dist.is_available() and dist.is_initialized()) else 'global'
        return torch.cuda.graph(graph, pool=pool, capture_error_mode=mode)

    def _pending_params(self):
        return None if self._pending is None else self._pending[1]

    def _finish_pending(self):
        """Complete a step whose optimiser part was deferred: wait for its gradient exchange, replay its optimiser graph."""
        if self._pending is None:
            return
        st, flat, optim, g2 = self._pending
        self._pending = None
        self.dp.wait(flat)
        g2.replay()
        self._after_optimizer(st, flat, optim)

    def _after_optimizer(self, st, flat, optim):
        optim.last_runs = st['runs']
        optim.note_replayed_step()
        # the replay updated the parameters through raw pointers: packed weights are stale (a host-side counter; the next
        # _run / fisher_sweep refreshes the pack group before it replays anything)
        op.bump_weights_epoch(flat.params)
        self.losses.update(st['losses'])                      # eager steps in between may have re-bound the entries

    def _graph_signature(self, optim):
        """What a captured step bakes in from the host: which parameters step (requires_grad pattern of the stage) and
        how they group into runs of equal step count (one bias-correction row per run).  A checkpoint resume or a
        stage change alters it; the step is then re-captured instead of replayed with stale grouping."""
        fp = optim.fp
        active = tuple(fp.params[i].requires_grad for i in fp.opt_idx)
        st = [optim.steps[i] for i in fp.opt_idx]
        rel = tuple(s - st[0] for s in st)
        # (+ whether the step gathers its latents from the pool: mapping-network parameters that become trainable after a
        # capture switch the step back to drawing and mapping its own noise)
        return (active, rel, optim.lr, tuple(optim.betas), optim.eps, self._pool_ok())

    def invalidate_graphs(self):
        """Drop every captured step (after loading a checkpoint or changing optimiser hyper-parameters)."""
        self._finish_pending()
        for st in self._gs.values():
            st.pop('graphs', None)
            st['n'] = 0                                       # two eager warm-up steps before the next capture
        self._fisher_state = None
        self.__dict__.pop('_lat_pool', None)                  # pooled W rows belong to the weights they were mapped with

    # ---- steps (each returns the loss tensor; no host sync)
    def d_step(self, real_img, noise, i=10 ** 9, g_noise=None, graph=False):
        self._set_d_stage(i)
        key = 'd' if graph else None
        batch = real_img.shape[0]

        def fb():
            with torch.no_grad():
                if graph:
                    fake_img, _ = self.g([self._graph_latents(key, batch)], input_is_latent=True, noise=g_noise)
                else:
                    fake_img, _ = self.g(noise, noise=g_noise)
            # one pass over cat(fake, real): identical to the reference's two calls (per-call minibatch-stddev
            # statistics are kept), half the launches and twice the GEMM rows per launch
            with self._sink():                   # conv weight gradients are added straight into the flat buffer
                pred, _ = self.d(torch.cat([fake_img, real_img], 0), calls=2)
                fake_pred, real_pred = pred.chunk(2, 0)
                d_loss = d_logistic_loss(real_pred, fake_pred)
                self._zero_grad(self.d_flat)
                # bias / noise-strength sums: one second-stage launch for the whole pass; (RICK_WGRAD_OVERLAP=1: sunk weight
                # gradients on a second stream next to the data-gradient chain — measured slower, off by default, op/conv.py)
                with op.deferred_sums(), op.wgrad_overlap():
                    d_loss.backward()
            self.losses.update(d=d_loss.detach(), real_score=real_pred.detach().mean(), fake_score=fake_pred.detach().mean())
        self._run(key, fb, self.d_flat, self.d_optim, pre=(lambda: self._draw_inject(key)) if graph else None)
        return self.losses['d']

    def r1_step(self, real_img, i=10 ** 9, graph=False):
        cfg = self.cfg
        self._set_d_stage(i)

        def fb():
            real = real_img.detach().requires_grad_(True)
            with op.second_order():
                real_pred, _ = self.d(real)
                real_pred = real_pred.view(real.size(0), -1).mean(dim=1).unsqueeze(1)
                r1_loss = d_r1_loss(real_pred, real)
                self._zero_grad(self.d_flat)
                (cfg.r1 / 2 * r1_loss * cfg.d_reg_every + 0 * real_pred[0]).backward()
            self.losses['r1'] = r1_loss.detach()
        self._run('r1' if graph else None, fb, self.d_flat, self.d_optim)
        return self.losses['r1']

    def g_step(self, noise, g_noise=None, graph=False):
        key = 'g' if graph else None
        batch = self.cfg.batch
        box = {}

        def head():                              # generator forward: reads G only (D's update may still be in flight)
            with self._sink():
                if graph:
                    box['fake'], _ = self.g([self._graph_latents(key, batch)], input_is_latent=True, noise=g_noise)
                else:
                    box['fake'], _ = self.g(noise, noise=g_noise)

        def fb():
            with self._sink(), self._d_frozen():
                fake_pred, _ = self.d(box.pop('fake'))
                g_loss = g_nonsaturating_loss(fake_pred)
                self._zero_grad(self.g_flat)
                with op.deferred_sums(), op.wgrad_overlap():
                    g_loss.backward()
            self.losses['g'] = g_loss.detach()
        self._run(key, fb, self.g_flat, self.g_optim, pre=(lambda: self._draw_inject(key)) if graph else None, fb_head=head)
        return self.losses['g']

    def _mixed_latents(self, noise):
        """W-space latents [B, n_latent, 512] for a list of 1 or 2 z tensors — Generator.forward's own mixing
        (random inject_index, model_probe_tune.py:555-560)."""
        n_latent = self.g.n_latent
        w = [self.g.style(z) for z in noise]
        if len(w) < 2:
            return w[0].unsqueeze(1).repeat(1, n_latent, 1)
        k = random.randint(1, n_latent - 1)
        return torch.cat([w[0].unsqueeze(1).repeat(1, k, 1), w[1].unsqueeze(1).repeat(1, n_latent - k, 1)], 1)

    def plr_step(self, noise, pl_noise=None, g_noise=None, graph=False):
        return self._plr_step(noise, pl_noise, g_noise, graph)

    def _plr_step(self, noise, pl_noise, g_noise, graph):
        cfg = self.cfg
        key = 'plr' if graph else None
        batch = max(1, cfg.batch // cfg.path_batch_shrink)
        if not torch.is_tensor(self.mean_path_length):      # persistent device scalar, updated in place (graph-safe)
            self.mean_path_length = torch.full((), float(self.mean_path_length), device=self.device)

        def fb():
            # The path-length gradient is taken w.r.t. the latents.  The optimiser owns no parameter of the mapping
            # network (train_dynamic_update_prune.py:908-917), so the latents enter as a detached leaf: the
            # reference's backward through the 8-layer MLP only produces gradients that are discarded, here it
            # (and its double backward) is not run.  Conv-weight gradients are unchanged.
            with torch.no_grad():
                lat = self._graph_latents(key, batch) if graph else self._mixed_latents(noise)
            lat = lat.detach().requires_grad_(True)
            with op.second_order():
                fake_img, latents = self.g([lat], input_is_latent=True, return_latents=True, noise=g_noise)
                path_loss, new_mean, path_lengths = g_path_regularize(fake_img, latents, self.mean_path_length,
                                                                      noise=pl_noise)
                self._zero_grad(self.g_flat)
                weighted = cfg.path_regularize * cfg.g_reg_every * path_loss
                if cfg.path_batch_shrink:
                    weighted = weighted + 0 * fake_img[0, 0, 0, 0]
                weighted.backward()
            self.mean_path_length.copy_(new_mean)
            self.losses.update(path=path_loss.detach(), path_length=path_lengths.detach().mean())
        self._run(key, fb, self.g_flat, self.g_optim, pre=(lambda: self._draw_inject(key)) if graph else None)
        return self.losses['path']

    def ema_step(self):
        """accumulate(g_ema, g), accumulate(d_ema, d) (:697-698) over ALL named parameters."""
        self._finish_pending()
        ema_flat(self.g_ema_flat, self.g_flat, self.cfg.ema_decay)
        ema_flat(self.d_ema_flat, self.d_flat, self.cfg.ema_decay)

    # ---- Fisher sweep (:214-393)
    def _fisher_sample(self, z, real, acc_g, acc_d, g_params, d_params, fixed_noise):
        """One sample of the sweep (:225-263): G / D forward at batch 1, both losses, grad^2 added to the accumulators."""
        fake, _ = self.g_ema([z.view(1, -1)], randomize_noise=not fixed_noise)
        fake_pred, _ = self.d_ema(fake)
        real_pred, _ = self.d_ema(real.view(1, 3, self.cfg.size, self.cfg.size))
        g_loss = g_nonsaturating_loss(fake_pred)
        d_loss = d_logistic_loss(real_pred, fake_pred)
        acc_g.add(autograd.grad(g_loss, g_params, retain_graph=True, allow_unused=True))
        acc_d.add(autograd.grad(d_loss, d_params, allow_unused=True))

    def fisher_sweep(self, latents, reals, first, fixed_noise=False):
        """latents: list of [1,512] tensors for THIS rank's samples; reals: matching [1,3,H,W].

        A sample is ~1 500 batch-1 launches issued through Python autograd — host-bound (21.5 ms per sample measured, of
        which the GPU works less than half).  With `use_graphs` the per-sample body is captured once (second sample of the
        first sweep; the first runs eagerly so that every packed-weight request and descriptor table exists) and replayed
        with the sample copied into static input buffers; the grad^2 accumulators are persistent tensors the captured
        launches add into.  Same kernels on the same data as the eager loop.

        Returns the two PERSISTENT accumulators (scaled grad^2 per parameter): the next sweep zeroes and refills them in
        place — clone what must outlive it."""
        cfg = self.cfg
        self._finish_pending()
        requires_grad(self.g_ema, True)
        requires_grad(self.d_ema, True)
        g_named, d_named = list(self.g_ema.named_parameters()), list(self.d_ema.named_parameters())
        g_params, d_params = [p for _, p in g_named], [p for _, p in d_named]
        st = self._fisher_state
        # what a captured per-sample graph bakes in: arithmetic mode, image size, which parameters take gradients
        sig = (fixed_noise, op.get_precision(), cfg.size, tuple(p.requires_grad for p in g_params + d_params))
        if st is None or st['sig'] != sig:
            st = self._fisher_state = {'sig': sig, 'acc': (FisherAccumulator(g_named), FisherAccumulator(d_named)),
                                       'z': torch.empty(1, cfg.latent, device=self.device),
                                       'real': torch.empty(1, 3, cfg.size, cfg.size, device=self.device), 'runs': 0, 'graph': None}
        acc_g, acc_d = st['acc']
        acc_g.zero_()
        acc_d.zero_()
        for z, real in zip(latents, reals):
            if not self.use_graphs:
                self._fisher_sample(z, real, acc_g, acc_d, g_params, d_params, fixed_noise)
                continue
            st['z'].copy_(z.view(1, -1))
            st['real'].copy_(real.view(1, 3, cfg.size, cfg.size))
            if st['graph'] is None and st['runs'] >= 1:
                for grp in self._ema_pack_groups:
                    grp.refresh()
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with self._capture(graph):
                    self._fisher_sample(st['z'], st['real'], acc_g, acc_d, g_params, d_params, fixed_noise)
                st['graph'] = graph
            if st['graph'] is not None:
                for grp in self._ema_pack_groups:         # the EMA weights move every iteration: repack on the host side
                    grp.refresh()
                st['graph'].replay()
            else:
                self._fisher_sample(st['z'], st['real'], acc_g, acc_d, g_params, d_params, fixed_noise)
            st['runs'] += 1
        scale = 1.0 / (cfg.num_fisher_img * cfg.batch)                 # :266-269
        acc_g.scale_(scale)
        acc_d.scale_(scale)
        conv, fc = g_filter_fim(acc_g.acc)
        dfim = d_filter_fim(acc_d.acc)
        if self.dp is not None:                                        # samples are sharded over ranks
            self.dp.all_reduce_vectors(list(conv.values()) + list(fc.values()) + list(dfim.values()))
        to_np = lambda d: {k: v.detach().cpu().numpy() for k, v in d.items()}   # noqa: E731
        self.fim = (to_np(conv), to_np(fc), to_np(dfim))
        self.idx_freeze_g, _, prune_g = decide_g(self.fim[0], self.fim[1], cfg.fisher_quantile, cfg.prune_quantile)
        self.idx_freeze_d, _, prune_d = decide_d(self.fim[2], cfg.fisher_quantile, cfg.prune_quantile)
        if first:
            self.zero_idx_g, self.zero_idx_d = prune_g, prune_d
        else:
            self.zero_idx_g = zero_idx_merge(self.zero_idx_g, prune_g)
            self.zero_idx_d = zero_idx_merge(self.zero_idx_d, prune_d)
        self.g_optim.set_mask(build_mask(self.g_flat, self.idx_freeze_g, self.zero_idx_g))
        self.d_optim.set_mask(build_mask(self.d_flat, self.idx_freeze_d, self.zero_idx_d))
        return acc_g, acc_d

    # ---- one iteration in the reference's order (:395-589, 697-698)
    def iteration(self, i, real_img, fisher_inputs=None):
        cfg = self.cfg
        if fisher_inputs is not None and i - cfg.warmup_iter >= 0 and (i - cfg.warmup_iter) % cfg.fisher_freq == 0:
            self.fisher_sweep(*fisher_inputs, first=(i == cfg.warmup_iter))
        graph = self.use_graphs and i >= cfg.warmup_iter      # the warm-up stage has its own requires_grad pattern
        if graph:
            if self._real is None:
                self._real = torch.empty_like(real_img)
            self._real.copy_(real_img)                        # captured launches read this buffer
            real_img = self._real
        nz = (lambda b: None) if graph else (lambda b: mixing_noise(b, cfg.latent, cfg.mixing, self.device))
        self._mark('begin')
        self.d_step(real_img, nz(cfg.batch), i, graph=graph)
        self._mark('d')
        if i % cfg.d_reg_every == 0:
            self.r1_step(real_img, i, graph=graph)
            self._mark('r1')
        if i >= cfg.warmup_iter:
            self.g_step(nz(cfg.batch), graph=graph)
            self._mark('g')
            if i % cfg.g_reg_every == 0:
                self.plr_step(nz(max(1, cfg.batch // cfg.path_batch_shrink)), graph=graph)
                self._mark('plr')
        self.ema_step()
        self._mark('ema')

    def _mark(self, name):
        """bench.py's per-step-type timing: a HIP event on the launch stream after each step (no host sync).  Under data
        parallelism with step graphs the optimiser part of a step is deferred behind the next step's head (`_run`), so the
        event of step k is recorded before its all-reduce wait and optimiser graph have run: that time is charged to step
        k + 1 — per-step times of `--gpus N` runs are pipelined and not comparable with single-GPU ones;
        `ms_per_step` and `value` are unaffected (whole iterations, fenced)."""
        if self.step_events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.step_events.append((name, e))
