"""Fused bias + (noise) + LeakyReLU on MI355X — drop-in for the reference's
``op.fused_leaky_relu`` / ``op.FusedLeakyReLU`` / ``op.FusedLeakyReLU_kml`` (op/fused_act.py:73-107).

    y = scale * leaky_relu(x + bias[c] (+ noise_weight * noise), negative_slope)

The gradient only needs the sign pattern of the OUTPUT (op/fused_act.py:22,28-30), so with
m = (y > 0 ? 1 : slope) the whole derivative tower is two mutually adjoint linear maps
    L (v, b, w) = scale * m * (v + b[c] + w * noise)       (rick_bias_act_f32, act=3, grad=1)
    L*(g)       = (scale*m*g, sum_rows(.), sum(. * noise))  (rick_bias_act_bwd_f32, one pass,
                                                             wavefront-shuffle reductions)
and every order of autograd alternates between them.  The optional noise term fuses
NoiseInjection (model_probe_tune.py:287-298) into the same pass; without it the op is exactly
the reference's.  Tensors are processed channels-last as [rows, C].
"""
import math

import torch
from torch import nn
from torch.autograd import Function

from .._lib import check, lib, ptr, require_cuda_f32, stream_ptr


def grad_sink_enabled():
    from .conv import grad_sink_enabled as f
    return f()


def param_like(t):
    from .conv import param_like as f
    return f(t)


def skip_param_grad(flag):
    from .conv import skip_param_grad as f
    return f(flag)


def _as_rows(x):
    """-> (tensor in [rows, C] memory order, rows, C, hw)."""
    if x.ndim == 2:
        return x.contiguous(), x.shape[0], x.shape[1], 1
    if x.ndim == 4:
        xc = x.contiguous(memory_format=torch.channels_last)
        hw = x.shape[2] * x.shape[3]
        return xc, x.shape[0] * hw, x.shape[1], hw
    raise RuntimeError(f'fused_leaky_relu: expected 2-D or 4-D input, got {x.ndim}-D')


def _noise_args(noise, x):
    if noise is None:
        return None, 1, 1
    if noise.ndim != 4 or noise.shape[1] != 1 or noise.shape[2:] != x.shape[2:] or noise.shape[0] not in (1, x.shape[0]):
        raise RuntimeError(f'noise must be [N or 1, 1, H, W] matching input {tuple(x.shape)}, got {tuple(noise.shape)}')
    return noise.contiguous(), noise.shape[0], noise.shape[2] * noise.shape[3]


def _pointwise(x, bias, ref, grad_mode, slope, scale, noise, nw):
    xr, rows, c, hw = _as_rows(x)
    out = torch.empty_like(xr)
    refr = None
    if ref is not None:
        refr, _, _, _ = _as_rows(ref)
    nz, nb, nhw = _noise_args(noise, x) if noise is not None else (None, 1, 1)
    from .conv import hbm_launch
    check(hbm_launch('bias_act', 4 * xr.numel() * (2 if refr is None else 3), lib.rick_bias_act_f32, ptr(xr), ptr(bias), ptr(refr),
                     ptr(out), xr.numel(), 1, c, 3, grad_mode, slope, scale, ptr(nz), ptr(nw), hw * c, c, nb, nhw, stream_ptr()),
          'rick_bias_act_f32')
    return out


def param_sink(p, numel, enabled):
    """The parameter's ``.grad`` when a [numel] gradient may be ADDED straight into it (op.grad_sink(), see op/conv.py),
    else None.  `enabled` is the switch as the op's forward saw it."""
    if not enabled or p is None:
        return None
    if not (p.is_leaf and p.requires_grad and p.grad is not None and p.grad.is_contiguous() and p.numel() == numel):
        return None
    return p.grad


# Deferred second stages (process-wide on purpose, like conv._param_grads_off: the backward passes that append run on the
# autograd engine's thread).  Inside `deferred_sums()` a bias / noise-strength gradient that goes into a gradient sink keeps
# its per-block partial rows and is summed, together with every other one of the step, by ONE launch when the context exits
# (rick_colsum_multi_f32: bit-identical item by item) — nobody reads those sums before the optimiser.  29 launches of ~4.7 us
# per train iteration become 2.
_deferred = None
_deferred_owner = None      # ident of the thread that opened the outermost context
_DEFER_OFF = bool(__import__('os').environ.get('RICK_NO_DEFER'))       # (A/B switch, tools/ab_env.sh)


class deferred_sums:
    """with op.deferred_sums(): loss.backward()

    ONE backward pass at a time, process-wide: the list is filled from the autograd engine's threads, which cannot be told
    apart by the thread that called backward().  A second thread that enters while a context is open gets a RuntimeError
    instead of another thread's sums (nn.DataParallel worker threads only run FORWARD passes; backward is one call from the
    main thread — INTEGRATION.md section 1).  Nesting on the opening thread is allowed."""

    def __enter__(self):
        global _deferred, _deferred_owner
        import threading
        me = threading.get_ident()
        if _deferred_owner is not None and _deferred_owner != me:
            raise RuntimeError('op.deferred_sums(): already open on another thread (one backward pass at a time)')
        self.prev, self.prev_owner = _deferred, _deferred_owner
        _deferred, _deferred_owner = (None if _DEFER_OFF else []), me
        return self

    def __exit__(self, *exc):
        global _deferred, _deferred_owner
        items, _deferred, _deferred_owner = _deferred, self.prev, self.prev_owner
        if exc[0] is None:
            flush_colsums(items)


def defer_colsum(part, out, nb, stride, ncols, col0=0, out2=None, split=0, accumulate=1):
    """Queue out[c] (+)= sum_r part[r * stride + col0 + c]; False when no deferred_sums() context is open.

    The items of one flush run as blocks of ONE launch, each doing a plain read-add-write of its destination: two items with the
    same destination would race.  A module applied twice in one backward (D called separately on real and fake, a shared bias)
    queues its sink twice — the items queued so far are then launched first (stream order = the order of the immediate path)."""
    global _deferred
    if _deferred is None:
        return False
    dst = {out.data_ptr()} | ({out2.data_ptr()} if out2 is not None else set())
    for it in _deferred:
        if it[1].data_ptr() in dst or (it[2] is not None and it[2].data_ptr() in dst):
            items = list(_deferred)
            del _deferred[:]
            flush_colsums(items)
            break
    _deferred.append((part, out, out2, int(nb), int(stride), int(ncols), int(col0), int(split), int(accumulate)))
    return True


def flush_colsums(items):
    if not items:
        return
    from .._lib import ColsumItem
    arr = (ColsumItem * len(items))()
    for a, (part, out, out2, nb, stride, ncols, col0, split, acc) in zip(arr, items):
        a.partials, a.out, a.out2 = part.data_ptr(), out.data_ptr(), (out2.data_ptr() if out2 is not None else None)
        a.nb, a.stride, a.ncols, a.col0, a.split, a.accumulate = nb, stride, ncols, col0, split, acc
    check(lib.rick_colsum_multi_f32(arr, len(items), stream_ptr()), 'rick_colsum_multi_f32')


def defer_act_sums(part, nblk, c, gb, gw):
    """The second stage of rick_bias_act_bwd_*'s partial rows [nblk][c + 1] into the sinks gb / gw, deferred."""
    if gb is not None and gw is not None:
        return defer_colsum(part, gb, nblk, c + 1, c + 1, 0, gw, c)
    if gb is not None:
        return defer_colsum(part, gb, nblk, c + 1, c, 0)
    return defer_colsum(part, gw, nblk, c + 1, 1, c)


def act_adjoint_dot(g, y, noise, slope, scale, want_b, want_w, sink_b, sink_w, bias, noise_w, d):
    """L* as in _ActAdjoint (first order only, no graph) that ALSO returns, from the same pass, the demodulation gradient of
    the modulated convolution the activation follows: gd[n, c] = sum_hw gx * conv_out / d[n, c] (conv_out reconstructed from y,
    rick_bias_act_bwd_dot_f32) — or None when the geometry has no fused form (the caller then takes the two-pass route).
    -> (gx, gb, gnw, gd)"""
    if g.ndim != 4 or noise is None and noise_w is not None:
        return None
    gr, rows, c, hw = _as_rows(g)
    if not lib.rick_bias_act_bwd_dot_ok(rows, c, hw):
        return None
    yr, _, _, _ = _as_rows(y)
    gx = torch.empty_like(gr)
    want_w = want_w and noise is not None
    nz, nb, nhw = _noise_args(noise, g) if noise is not None else (None, 1, 1)
    if (want_b and sink_b is None) or (want_w and sink_w is None):
        sink_b = sink_w = None
    sunk = (want_b and sink_b is not None) or (want_w and sink_w is not None)
    gb = (sink_b if sunk else torch.empty(c, device=g.device, dtype=g.dtype)) if want_b else None
    gw = (sink_w if sunk else torch.empty(1, device=g.device, dtype=g.dtype)) if want_w else None
    nblk = lib.rick_bias_act_bwd_blocks(rows, c)
    part = torch.empty(nblk * (c + 1), device=g.device, dtype=g.dtype) if (want_b or want_w) else None
    later = bool(part is not None and sunk and defer_act_sums(part, nblk, c, gb, gw))
    dpart = torch.empty(nblk * c, device=g.device, dtype=g.dtype)
    dc = d.contiguous()
    gd = torch.empty_like(dc)
    from .conv import hbm_launch
    check(hbm_launch('bias_act_bwd', 12 * gr.numel(), lib.rick_bias_act_bwd_dot_f32, ptr(gr), ptr(yr), ptr(gx), ptr(gb), ptr(gw), ptr(nz),
                     rows, c, hw, nb, nhw, slope, scale, ptr(part), int(sunk) | (2 if later else 0),
                     ptr(bias.contiguous() if bias is not None else None), ptr(noise_w.contiguous() if noise_w is not None else None),
                     ptr(dc), ptr(gd), ptr(dpart), stream_ptr()), 'rick_bias_act_bwd_dot_f32')
    stats['adjoint_dot'] += 1
    return gx, (None if sunk else gb), (None if sunk else gw), gd


stats = {'adjoint_dot': 0}      # launches of the fused adjoint + demodulation-gradient pass (tests)


class _ActAdjoint(Function):
    """L*: g -> (gx, gb, gnw) given the saved output y (and noise).  A gradient that is not wanted is None (not a zero
    tensor).  sink_b / sink_w: the parameters' .grad buffers — the reduction's second stage adds the sums into them and
    the corresponding output is None (nothing left for autograd to accumulate)."""

    @staticmethod
    def forward(ctx, g, y, noise, slope, scale, want_b, want_w, sink_b=None, sink_w=None):
        gr, rows, c, hw = _as_rows(g)
        yr, _, _, _ = _as_rows(y)
        gx = torch.empty_like(gr)
        want_w = want_w and noise is not None
        nz, nb, nhw = _noise_args(noise, g) if want_w else (None, 1, 1)
        if (want_b and sink_b is None) or (want_w and sink_w is None):
            sink_b = sink_w = None                     # one accumulate switch serves both sums
        sunk = (want_b and sink_b is not None) or (want_w and sink_w is not None)
        gb = (sink_b if sunk else torch.empty(c, device=g.device, dtype=g.dtype)) if want_b else None
        gw = (sink_w if sunk else torch.empty(1, device=g.device, dtype=g.dtype)) if want_w else None
        part = None
        later = False
        if want_b or want_w:
            nblk = lib.rick_bias_act_bwd_blocks(rows, c)
            part = torch.empty(nblk * (c + 1), device=g.device, dtype=g.dtype)
            later = sunk and defer_act_sums(part, nblk, c, gb, gw)       # second stage with the rest of the step's (deferred_sums)
        from .conv import hbm_launch
        check(hbm_launch('bias_act_bwd', 12 * gr.numel(), lib.rick_bias_act_bwd_f32, ptr(gr), ptr(yr), ptr(gx), ptr(gb), ptr(gw), ptr(nz),
                         rows, c, hw, nb, nhw, slope, scale, ptr(part), int(sunk) | (2 if later else 0), stream_ptr()),
              'rick_bias_act_bwd_f32')
        ctx.save_for_backward(y, noise)
        ctx.cfg = (slope, scale)
        if sunk:
            return gx, None, None
        return gx, gb, gw

    @staticmethod
    def backward(ctx, ggx, ggb, ggw):
        y, noise = ctx.saved_tensors
        slope, scale = ctx.cfg
        out = _ActLinear.apply(ggx, ggb, ggw, y, noise, slope, scale)
        return out, None, None, None, None, None, None, None, None


class _ActLinear(Function):
    """L: (v, b, w) -> scale * m(y) * (v + b[c] + w*noise)."""

    @staticmethod
    def forward(ctx, v, b, w, y, noise, slope, scale):
        ctx.save_for_backward(y, noise)
        ctx.cfg = (slope, scale)
        ctx.need = (b is not None, w is not None and noise is not None)
        nz = noise if (w is not None and noise is not None) else None
        return _pointwise(v, b.contiguous() if b is not None else None, y, 1, slope, scale, nz,
                          w.contiguous() if nz is not None else None)

    @staticmethod
    def backward(ctx, g):
        y, noise = ctx.saved_tensors
        slope, scale = ctx.cfg
        gx, gb, gw = _ActAdjoint.apply(g, y, noise, slope, scale, ctx.need[0], ctx.need[1])
        return gx, (gb if ctx.need[0] else None), (gw if ctx.need[1] else None), None, None, None, None


class _Act(Function):
    @staticmethod
    def forward(ctx, x, bias, noise, nw, slope, scale):
        y = _pointwise(x, bias.contiguous() if bias is not None else None, None, 0, slope, scale, noise,
                       nw.contiguous() if noise is not None else None)
        ctx.save_for_backward(y, noise)
        ctx.cfg = (slope, scale)
        ctx.need = (bias is not None, noise is not None)
        ctx.params = (bias, nw, grad_sink_enabled())      # the switch as the forward's thread sees it (op/conv.py)
        ctx.param_like = (param_like(bias), param_like(nw))
        return y

    @staticmethod
    def backward(ctx, g):
        y, noise = ctx.saved_tensors
        slope, scale = ctx.cfg
        bias, nw, sink = ctx.params
        want_b = ctx.need[0] and ctx.needs_input_grad[1] and not skip_param_grad(ctx.param_like[0])
        want_w = ctx.need[1] and ctx.needs_input_grad[3] and not skip_param_grad(ctx.param_like[1])
        sink = sink and not torch.is_grad_enabled()       # a twice-differentiable backward keeps its gradients in the graph
        gx, gb, gw = _ActAdjoint.apply(g, y, noise, slope, scale, want_b, want_w,
                                       param_sink(bias, y.shape[1], sink and want_b), param_sink(nw, 1, sink and want_w))
        return (gx, gb, None, gw, None, None)


def _any_pointwise(x, bias, ref, grad_mode, slope, scale):
    """fused_bias_act in float64 / float16 (rick_bias_act_any): x of any shape with channels on dim 1 (op/fused_act.py:52-57)."""
    from .._lib import DTYPE_CODE
    xc = x.contiguous()
    out = torch.empty_like(xc)
    step_b = 1
    for d in xc.shape[2:]:
        step_b *= d
    check(lib.rick_bias_act_any(ptr(xc), ptr(bias.contiguous() if bias is not None else None), ptr(ref.contiguous() if ref is not None else None),
                                ptr(out), DTYPE_CODE[x.dtype], xc.numel(), step_b, xc.shape[1] if bias is not None else 1, 3, grad_mode,
                                slope, scale, stream_ptr()), 'rick_bias_act_any')
    return out


class _ActAnyBackward(Function):
    """op/fused_act.py:19-48 (FusedLeakyReLUFunctionBackward) for float64 / float16."""

    @staticmethod
    def forward(ctx, g, out, slope, scale, has_bias):
        ctx.save_for_backward(out)
        ctx.cfg = (slope, scale, has_bias)
        gx = _any_pointwise(g, None, out, 1, slope, scale)
        gb = gx.sum([0] + list(range(2, gx.ndim))).detach() if has_bias else None
        return gx, gb

    @staticmethod
    def backward(ctx, ggx, ggb):
        (out,) = ctx.saved_tensors
        slope, scale, has_bias = ctx.cfg
        if ggx is None:
            ggx = torch.zeros_like(out)
        return _any_pointwise(ggx, ggb if has_bias else None, out, 1, slope, scale), None, None, None, None


class _ActAny(Function):
    """op/fused_act.py:51-70 (FusedLeakyReLUFunction) for float64 / float16: same three-level structure as the reference."""

    @staticmethod
    def forward(ctx, x, bias, slope, scale):
        out = _any_pointwise(x, bias, None, 0, slope, scale)
        ctx.save_for_backward(out)
        ctx.cfg = (slope, scale, bias is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        slope, scale, has_bias = ctx.cfg
        gx, gb = _ActAnyBackward.apply(g, out, slope, scale, has_bias)
        return gx, gb, None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    """Reference signature (op/fused_act.py:106-107); float16 / float32 / float64 like the reference extension."""
    from .._lib import require_cuda_float
    if require_cuda_float(input, bias) != torch.float32:
        return _ActAny.apply(input, bias, float(negative_slope), float(scale))
    return _Act.apply(input, bias, None, None, float(negative_slope), float(scale))


def fused_noise_bias_act(input, bias, noise, noise_weight, negative_slope=0.2, scale=2 ** 0.5):
    """activate(noise_injection(input)) of StyledConv (model_probe_tune.py:343-346) in one pass.
    noise: [N or 1, 1, H, W]; noise_weight: Parameter[1]."""
    require_cuda_f32(input, bias, noise, noise_weight)
    return _Act.apply(input, bias, noise, noise_weight, float(negative_slope), float(scale))


class FusedLeakyReLU(nn.Module):
    """Same constructor and state (``bias``) as the reference module (op/fused_act.py:73-82)."""

    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


class FusedLeakyReLU_kml(nn.Module):
    """op/fused_act.py:85-103 (extra ``b_vector``; unused by the training script)."""

    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.b_vector = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        bias = self.bias + self.b_vector if self.b_vector.requires_grad else self.bias
        return fused_leaky_relu(input, bias, self.negative_slope, self.scale)
