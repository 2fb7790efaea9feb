"""HBM-bound helper ops of the hot path (rick_amd/csrc/elementwise.hip, thin.hip), each wired
into autograd as a family closed under differentiation (needed by R1 / path-length terms).

    chan_scale(x, s)   x[n,c,h,w] * s[n,c]           (style modulation / demodulation,
    hw_dot(a, b)       sum_hw a*b -> [n,c]             model_probe_tune.py:246-251)
    add_scale(a, b, k) (a + b) * k                    (ResBlock merge, model_probe_tune.py:658)
    thin_fwd / thin_bwdx / thin_wgrad                 (3-channel 1x1 products: ToRGB :351-370,
                                                       discriminator input conv :679)
    minibatch_stddev                                  (model_probe_tune.py:748-756)
"""
import torch
from torch.autograd import Function

from .._lib import check, lib, ptr, require_cuda_f32, stream_ptr


stats = {'fork_inplace': 0, 'fork_copy': 0}     # how op.torgb_fork's backward added the branch gradient (tests)


def _nhwc(x):
    return x.contiguous(memory_format=torch.channels_last)


# ------------------------------------------------------------------ chan_scale / hw_dot
def _chan_scale_raw(x, s):
    x = _nhwc(x)
    s = s.contiguous()
    n, c, h, w = x.shape
    if s.shape != (n, c):
        raise RuntimeError(f'chan_scale: scale must be [{n},{c}], got {tuple(s.shape)}')
    y = torch.empty_like(x)
    check(lib.rick_chan_scale_f32(ptr(x), ptr(s), ptr(y), n, h * w, c, stream_ptr()), 'rick_chan_scale_f32')
    return y


def _hw_dot_raw(a, b, divisor=None):
    """sum_hw a*b -> [n, c]; `divisor` [n, c]: the sums are divided by it in the reduction's second stage."""
    a, b = _nhwc(a), _nhwc(b)
    n, c, h, w = a.shape
    d = torch.empty((n, c), device=a.device, dtype=a.dtype)
    nb = lib.rick_hw_dot_blocks(h * w)
    part = torch.empty(nb * n * c, device=a.device, dtype=a.dtype)
    check(lib.rick_hw_dot_f32(ptr(a), ptr(b), ptr(d), n, h * w, c, ptr(part), ptr(divisor), stream_ptr()), 'rick_hw_dot_f32')
    return d


class _ChanScale(Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.save_for_backward(x, s)
        return _chan_scale_raw(x, s)

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        gx = _ChanScale.apply(g, s) if ctx.needs_input_grad[0] else None
        gs = _HwDot.apply(g, x) if ctx.needs_input_grad[1] else None
        return gx, gs


class _HwDot(Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return _hw_dot_raw(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = _ChanScale.apply(b, g) if ctx.needs_input_grad[0] else None
        gb = _ChanScale.apply(a, g) if ctx.needs_input_grad[1] else None
        return ga, gb


def chan_scale(x, s):
    require_cuda_f32(x, s)
    return _ChanScale.apply(x, s)


def hw_dot(a, b):
    require_cuda_f32(a, b)
    return _HwDot.apply(a, b)


# --------------------------------------------------------------------------- add_scale
class _AddScale(Function):
    @staticmethod
    def forward(ctx, a, b, k):
        ctx.k = k
        a = _nhwc(a) if a.ndim == 4 else a.contiguous()
        if b is not None:
            b = (_nhwc(b) if b.ndim == 4 else b.contiguous())
            if b.shape != a.shape:
                raise RuntimeError('add_scale: shape mismatch')
        y = torch.empty_like(a)
        check(lib.rick_add_scale_f32(ptr(a), ptr(b), ptr(y), a.numel(), k, stream_ptr()), 'rick_add_scale_f32')
        return y

    @staticmethod
    def backward(ctx, g):
        gs = _AddScale.apply(g, None, ctx.k)
        return gs, (gs if ctx.needs_input_grad[1] else None), None


def add_scale(a, b, k):
    require_cuda_f32(a, b)
    return _AddScale.apply(a, b, float(k))


# ------------------------------------------------------------------------- thin products
def _thin_shapes(x, J):
    n, c, h, w = x.shape
    if c % 4:
        raise RuntimeError('thin ops need C % 4 == 0')
    if not 1 <= J <= 4:
        raise RuntimeError('thin ops support 1..4 thin channels')
    return n, c, h, w


def _wb(W, n):
    """W: [N or 1, J, C] -> (contiguous tensor, batch stride)."""
    W = W.contiguous()
    if W.shape[0] not in (1, n):
        raise RuntimeError('thin ops: W batch must be 1 or N')
    return W, (0 if W.shape[0] == 1 else W.shape[1] * W.shape[2])


class _ThinFwd(Function):
    """t[n,j,h,w] = sum_c x[n,c,h,w] W[n,j,c]   (x channels-last, t planar)."""

    @staticmethod
    def forward(ctx, x, W):
        ctx.save_for_backward(x, W)          # the ORIGINAL inputs, so higher-order graphs reach them
        x = _nhwc(x)
        n, c, h, w = _thin_shapes(x, W.shape[1])
        Wc, bs = _wb(W, n)
        t = torch.empty((n, W.shape[1], h, w), device=x.device, dtype=x.dtype)
        from .conv import hbm_launch
        check(hbm_launch('thin', 4 * (x.numel() + t.numel()), lib.rick_thin_fwd_f32, ptr(x), ptr(Wc), bs, None, ptr(t), n, h * w, c,
                         W.shape[1], stream_ptr()), 'rick_thin_fwd_f32')
        return t

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        gx = _ThinBwdX.apply(g, W) if ctx.needs_input_grad[0] else None
        gW = None
        if ctx.needs_input_grad[1]:
            gW = _ThinWgrad.apply(g, x)
            if W.shape[0] == 1:
                gW = gW.sum(0, keepdim=True)
        return gx, gW


class _ThinBwdX(Function):
    """x[n,c,h,w] = sum_j t[n,j,h,w] W[n,j,c]   (t planar, x channels-last)."""

    @staticmethod
    def forward(ctx, t, W):
        ctx.save_for_backward(t, W)
        t = t.contiguous()
        n, J, h, w = t.shape
        c = W.shape[2]
        Wc, bs = _wb(W, n)
        x = torch.empty((n, c, h, w), device=t.device, dtype=t.dtype, memory_format=torch.channels_last)
        _thin_shapes(x, J)
        from .conv import hbm_launch
        check(hbm_launch('thin', 4 * (x.numel() + t.numel()), lib.rick_thin_bwdx_f32, ptr(t), ptr(Wc), bs, ptr(x), n, h * w, c, J,
                         stream_ptr()), 'rick_thin_bwdx_f32')
        return x

    @staticmethod
    def backward(ctx, gg):
        t, W = ctx.saved_tensors
        gt = _ThinFwd.apply(gg, W) if ctx.needs_input_grad[0] else None
        gW = None
        if ctx.needs_input_grad[1]:
            gW = _ThinWgrad.apply(t, gg)
            if W.shape[0] == 1:
                gW = gW.sum(0, keepdim=True)
        return gt, gW


class _ThinWgrad(Function):
    """G[n,j,c] = sum_hw t[n,j,h,w] x[n,c,h,w]."""

    @staticmethod
    def forward(ctx, t, x):
        ctx.save_for_backward(t, x)
        t = t.contiguous()
        x = _nhwc(x)
        n, c, h, w = _thin_shapes(x, t.shape[1])
        J = t.shape[1]
        G = torch.empty((n, J, c), device=x.device, dtype=x.dtype)
        nb = lib.rick_thin_wgrad_blocks(h * w)
        part = torch.empty(nb * n * J * c, device=x.device, dtype=x.dtype)
        check(lib.rick_thin_wgrad_f32(ptr(t), ptr(x), ptr(G), n, h * w, c, J, ptr(part), stream_ptr()),
              'rick_thin_wgrad_f32')
        return G

    @staticmethod
    def backward(ctx, gG):
        t, x = ctx.saved_tensors
        gt = _ThinFwd.apply(x, gG) if ctx.needs_input_grad[0] else None
        gx = _ThinBwdX.apply(t, gG) if ctx.needs_input_grad[1] else None
        return gt, gx


def _torgb_param_grads(ctx, g, x, w, s):
    gw = gs = gb = None
    if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
        G = _ThinWgrad.apply(g, x)                                  # [n, J, c] = d loss / d W[n]
        if ctx.needs_input_grad[1]:
            gw = ctx.wscale * torch.einsum('njc,nc->jc', G, s)
        if ctx.needs_input_grad[2]:
            gs = torch.einsum('njc,jc->nc', G, ctx.wscale * w)
    if ctx.bias_shape is not None and ctx.needs_input_grad[3]:
        gb = g.sum((0, 2, 3)).view(ctx.bias_shape)
    return gw, gs, gb


class _ToRGB(Function):
    """rgb = ToRGB's modulated 1x1 conv + bias (+ upsampled skip) in ONE launch (model_probe_tune.py:246-248, 366-370):
    the per-sample weight (scale * w) * s is formed inside the kernel.  First-order autograd; the data gradient is one
    launch too, the weight / style / bias gradients (the training steps never need them: the optimiser owns no ToRGB
    parameter, train_dynamic_update_prune.py:908-917) are composed from thin_wgrad."""

    @staticmethod
    def forward(ctx, x, w, s, bias, add, wscale):
        xc = _nhwc(x)
        n, c, h, wd = _thin_shapes(xc, w.shape[0])
        J = w.shape[0]
        wc, sc = w.contiguous(), s.contiguous()
        bc = bias.reshape(-1).contiguous() if bias is not None else None
        ac = add.contiguous() if add is not None else None
        if sc.shape != (n, c) or wc.shape != (J, c) or (ac is not None and ac.shape != (n, J, h, wd)):
            raise RuntimeError('torgb: shape mismatch')
        t = torch.empty((n, J, h, wd), device=x.device, dtype=x.dtype)
        from .conv import hbm_launch
        check(hbm_launch('thin', 4 * (xc.numel() + t.numel() * (2 if ac is not None else 1)), lib.rick_torgb_fwd_f32, ptr(xc), ptr(wc),
                         ptr(sc), wscale, ptr(bc), ptr(ac), ptr(t), n, h * wd, c, J, stream_ptr()), 'rick_torgb_fwd_f32')
        ctx.save_for_backward(x, w, s)
        ctx.wscale, ctx.bias_shape = wscale, (bias.shape if bias is not None else None)
        ctx.bias_add = (bias, add)      # (only used by the create_graph route; bias is a Parameter, add the upsampled skip)
        return t

    @staticmethod
    def backward(ctx, g):
        x, w, s = ctx.saved_tensors
        if torch.is_grad_enabled():     # create_graph=True: per-sample weights + the thin product (closed under differentiation)
            from ._twice import second_order_backward
            bias, add = ctx.bias_add
            J, c = w.shape

            def compose(x_, w_, s_, b_, add_):
                out = _ThinFwd.apply(x_, (ctx.wscale * w_.view(1, J, c)) * s_.unsqueeze(1))
                if b_ is not None:
                    out = out + b_.view(1, J, 1, 1)
                return out if add_ is None else out + add_
            res = second_order_backward(compose, (x, w, s, bias, add), ctx.needs_input_grad[:5], g)
            return (*res, None)
        g = g.contiguous()
        n, J, h, wd = g.shape
        c = w.shape[1]
        gx = gw = gs = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty((n, c, h, wd), device=g.device, dtype=g.dtype, memory_format=torch.channels_last)
            from .conv import hbm_launch
            check(hbm_launch('thin', 4 * (gx.numel() + g.numel()), lib.rick_torgb_bwdx_f32, ptr(g), ptr(w.contiguous()),
                             ptr(s.contiguous()), ctx.wscale, ptr(gx), n, h * wd, c, J, stream_ptr()), 'rick_torgb_bwdx_f32')
        gw, gs, gb = _torgb_param_grads(ctx, g, x, w, s)
        return gx, gw, gs, gb, (g if ctx.needs_input_grad[4] else None), None


class _ToRGBFork(Function):
    """(x, rgb) = (x, ToRGB(x)): the activation that feeds ToRGB also feeds the next layer (model_probe_tune.py:362-368).  As one
    node with two outputs, the node receives BOTH gradients of the branch point and the ToRGB data gradient is added into the
    next layer's data gradient by the launch that produces it (rick_torgb_bwdx_acc_f32) — instead of a full-size tensor written
    by ToRGB's backward, read again and added by autograd (three passes over the activation; same fp32 addition: torch.equal)."""

    @staticmethod
    def forward(ctx, x, w, s, bias, add, wscale):
        t = _ToRGB.forward(ctx, x, w, s, bias, add, wscale)
        return x.view_as(x), t

    @staticmethod
    def backward(ctx, gx_next, g):
        if g is None:
            return gx_next, None, None, None, None, None
        if gx_next is None or not ctx.needs_input_grad[0] or torch.is_grad_enabled():
            res = list(_ToRGB.backward(ctx, g))         # (create_graph=True: the twice-differentiable composition)
            if gx_next is not None and ctx.needs_input_grad[0]:
                res[0] = gx_next if res[0] is None else res[0] + gx_next
            return tuple(res)
        x, w, s = ctx.saved_tensors
        g = g.contiguous()
        n, J, h, wd = g.shape
        c = w.shape[1]
        gw, gs, gb = _torgb_param_grads(ctx, g, x, w, s)
        # In place only into a buffer its producer marked as exclusively owned (autograd's contract forbids modifying a
        # gradient input that somebody else may hold: retain_grad() on the forked activation, a backward that returns one
        # tensor for two inputs — neither carries the mark).  The mark is consumed; anything else is copied first (ADVICE
        # round 4).  NOT covered: a tensor hook registered on the forked activation receives this same marked
        # tensor before the node runs; a hook that KEEPS its argument (instead of reading or replacing it) then sees the ToRGB
        # contribution added into it.  Nothing in the package registers such a hook (rick_amd.dist hooks parameters, whose
        # gradients never pass here); a caller who does must clone in the hook.
        ok = (gx_next.__dict__.pop('_rick_owned', False) and gx_next.dtype == torch.float32
              and tuple(gx_next.shape) == (n, c, h, wd) and gx_next.data_ptr() % 16 == 0
              and gx_next.is_contiguous(memory_format=torch.channels_last))
        gx = gx_next if ok else gx_next.contiguous(memory_format=torch.channels_last).clone()
        for a in ('_rick_split', '_rick_amax', '_rick_bound'):      # hand-over attributes described gx_next alone, not the sum
            gx.__dict__.pop(a, None)
        stats['fork_inplace' if ok else 'fork_copy'] += 1
        from .conv import hbm_launch
        check(hbm_launch('thin', 4 * (2 * gx.numel() + g.numel()), lib.rick_torgb_bwdx_acc_f32, ptr(g), ptr(w.contiguous()),
                         ptr(s.contiguous()), ctx.wscale, ptr(gx), n, h * wd, c, J, stream_ptr()), 'rick_torgb_bwdx_acc_f32')
        return gx, gw, gs, gb, (g if ctx.needs_input_grad[4] else None), None


def torgb_fork(x, w, s, bias=None, add=None, wscale=1.0):
    """(x', rgb): rgb = torgb(x, ...), x' = x for the next layer — see _ToRGBFork."""
    require_cuda_f32(x, w, s, bias, add)
    xo, t = _ToRGBFork.apply(x, w, s, bias, add, float(wscale))
    from . import split as sp
    sp.rehand(x, xo)      # still-valid hand-over attributes of the producing layer (op/split.py) travel with x
    return xo, t


def torgb(x, w, s, bias=None, add=None, wscale=1.0):
    """sum_c x[n,c,h,w] * (wscale * w[j,c]) * s[n,c] + bias[j] + add[n,j,h,w]  -> planar [N, J, H, W]."""
    require_cuda_f32(x, w, s, bias, add)
    return _ToRGB.apply(x, w, s, bias, add, float(wscale))


def thin_fwd(x, W):
    require_cuda_f32(x, W)
    return _ThinFwd.apply(x, W)


def thin_bwdx(t, W):
    require_cuda_f32(t, W)
    return _ThinBwdX.apply(t, W)


# ------------------------------------------------------------------- minibatch stddev
class _MbStd(Function):
    """HIP path with a hand-written first-order backward; under create_graph=True the backward differentiates the
    composite below (op/_twice.py)."""

    @staticmethod
    def forward(ctx, x_in, groups, stddev=(25, 1)):
        x = _nhwc(x_in)
        b, c, h, w = x.shape
        out = torch.empty((b, c + 1, h, w), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
        stat = torch.empty(groups, device=x.device, dtype=x.dtype)
        check(lib.rick_mbstd_fwd_f32(ptr(x), ptr(out), ptr(stat), b, h * w, c, groups, stream_ptr()), 'rick_mbstd_fwd_f32')
        ctx.save_for_backward(x if x is x_in else x_in)
        ctx.groups, ctx.stddev = groups, stddev
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        if torch.is_grad_enabled():     # create_graph=True (R1): the formula composed from tensor ops, per concatenated call
            from ._twice import second_order_backward
            calls, (sg, sf) = ctx.groups, ctx.stddev
            (gx,) = second_order_backward(
                lambda x_: (_mbstd_composite(x_, sg, sf) if calls == 1
                            else torch.cat([_mbstd_composite(xc, sg, sf) for xc in x_.chunk(calls)], 0)), (x,), (True,), g)
            return gx, None, None
        g = _nhwc(g)
        x = _nhwc(x)
        b, c, h, w = x.shape
        gx = torch.empty_like(x)
        check(lib.rick_mbstd_bwd_f32(ptr(x), ptr(g), ptr(gx), b, h * w, c, ctx.groups, stream_ptr()), 'rick_mbstd_bwd_f32')
        return gx, None, None


def _mbstd_composite(x, stddev_group, stddev_feat):
    b, c, h, w = x.shape
    group = min(b, stddev_group)
    s = x.reshape(group, -1, stddev_feat, c // stddev_feat, h, w)
    s = torch.sqrt(s.var(0, unbiased=False) + 1e-8)
    s = s.mean([2, 3, 4], keepdim=True).squeeze(2)
    s = s.repeat(group, 1, h, w)
    return torch.cat([x, s], 1)


def minibatch_stddev(x, stddev_group=25, stddev_feat=1, second_order=False, calls=1):
    """cat([x, stddev channel]) as in model_probe_tune.py:748-756.  `calls` > 1 means the batch is the
    concatenation of that many independent discriminator calls (each keeps its own statistics, exactly as
    if the module had been called once per chunk).  The HIP kernel covers the training configuration
    (per-call batch <= stddev_group, stddev_feat == 1); other shapes and double-differentiable calls (R1) use
    the same formula composed from device tensor ops on this [B,512,4,4] tensor (8 K elements per sample)."""
    require_cuda_f32(x)
    b = x.shape[0]
    if b % calls:
        raise RuntimeError('minibatch_stddev: batch not divisible by the number of concatenated calls')
    per = b // calls
    if per <= stddev_group and stddev_feat == 1 and not second_order:
        return _MbStd.apply(x, calls, (stddev_group, stddev_feat))
    if calls == 1:
        return _mbstd_composite(x, stddev_group, stddev_feat)
    return torch.cat([_mbstd_composite(xc, stddev_group, stddev_feat) for xc in x.chunk(calls)], 0)


# ------------------------------------------------------------------------- short-batch EqualLinear
def equal_linear(x, weight, bias, scale, lr_mul=1.0, activate=False, pixelnorm=False, negative_slope=0.2, gain=2 ** 0.5):
    """act(scale * PixelNorm?(x) @ weight^T + bias * lr_mul) for x [B <= 16, K] in one launch (EqualLinear forward,
    model_probe_tune.py:157-168; act = fused_leaky_relu's gain * leaky_relu).  No autograd graph is recorded."""
    require_cuda_f32(x, weight, bias)
    x = x.detach().contiguous()
    w = weight.detach().contiguous()
    B, K = x.shape
    O = w.shape[0]
    if w.shape[1] != K:
        raise RuntimeError(f'equal_linear: input has {K} features, weight expects {w.shape[1]}')
    out = torch.empty((B, O), device=x.device, dtype=x.dtype)
    b = bias.detach().contiguous() if bias is not None else None
    check(lib.rick_equal_linear_f32(ptr(x), ptr(w), ptr(b), ptr(out), B, K, O, float(scale), float(lr_mul), int(activate),
                                    float(negative_slope), float(gain), int(pixelnorm), stream_ptr()), 'rick_equal_linear_f32')
    return out
