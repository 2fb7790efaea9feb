"""Modulated convolution (StyleGAN2 weight modulation / demodulation,
model_probe_tune.py:188-284) without per-sample weights.

The reference builds W'[b,o,i,k] = scale*W[o,i,k]*s[b,i]*d[b,o] and runs a grouped conv with
B groups.  Algebraically  y[b,o] = d[b,o] * sum_{i,k} scale*W[o,i,k] * (s[b,i] x[b,i]),
with d = rsqrt(sum_{i,k} (scale*W*s)^2 + 1e-8) = rsqrt(s^2 @ Wsq^T + 1e-8): one shared-weight
convolution over the whole batch (GEMM M = B*H*W) with the modulation folded into the operand
load (iscale) and the demodulation into the epilogue (oscale) of the MFMA kernel.

Two implementations with identical values:
  * `modulated_conv_fused`   one kernel per direction; hand-written first-order backward, and under create_graph=True a
                              backward that differentiates the composed form below (op/_twice.py)
  * `modulated_conv_composed` chan_scale -> conv -> chan_scale from the closed primitive
                              family, differentiable to any order (used under op.second_order()).
"""
import ctypes
import os
import math

import numpy as np
import torch
from torch.autograd import Function

from .._lib import check, lib, ptr, require_cuda_f32, stream_ptr
from .conv import (_conv_launch, _convT_launch, _epilogue, _pack, _sink_target, _wgrad_launch, conv2d, conv_transpose2d,
                   grad_sink_enabled, param_like)
from .misc import _chan_scale_raw, _hw_dot_raw, chan_scale


# Split-image hand-over between the generator's fused layers: built, parity-tested (tests/test_gpu_split.py) and OFF by default.
# Unlike the discriminator's blocks (op/dblock.py), where the images REPLACE fp32 tensors nobody else reads and the exact maxima
# fall out of the conv epilogues, every generator activation is also read as fp32 (ToRGB, style and demodulation gradients), so
# the images are extra writes, the gradient maxima need stand-alone passes and each bound a tiny launch: same-box A/B 153.3 vs
# 154.5 images/s (-0.8 %): the MFMA kernels gained 6.4 ms over 30 iterations (7-9 % on the launches that moved, not the 11-26 %
# of the micro-benchmark on random data), the producers / maxima / bound launches cost 13.5 ms.  RICK_GSPLIT=1 switches it on.
_USE_SPLIT = bool(__import__('os').environ.get('RICK_GSPLIT'))


_FUSE_ADJOINT_DOT = not os.environ.get('RICK_NO_ADJOINT_DOT')      # (A/B switch)
stats = {'fprop': 0, 'dgrad': 0, 'wgrad': 0, 'produced': 0}     # launches that consumed / produced a split image (tests)


def _bound_tail(w0, c0, gain, nw=None, w_noise=None, bias=None, mul=None):
    """amax word  max|mul| * gain * (c0 * amax(w0) + |nw| * amax(w_noise) + max|bias|)  evaluated on the device."""
    from . import split as sp
    out = sp.new_amax(w0.device)
    check(lib.rick_bound_tail_f32(ptr(out), ptr(w0), float(c0), ptr(nw), ptr(w_noise), ptr(bias), bias.numel() if bias is not None else 0,
                                  float(gain), ptr(mul), mul.numel() if mul is not None else 0, stream_ptr()), 'rick_bound_tail_f32')
    return out


def _tensor_amax(t):
    """The running-maximum word of a tensor: the one its producer measured (attribute), else a stand-alone pass."""
    from . import split as sp
    a = sp.taken(t, '_rick_amax')
    return a if a is not None else sp.amax(t)


def _same_tensor(a, b):
    return a is b or (a is not None and b is not None and a.data_ptr() == b.data_ptr() and a.shape == b.shape
                      and a._version == b._version)


def _input_image(x, s):
    """The split image of s * x its producer attached to x, if it was made for THIS scale tensor."""
    from . import split as sp
    img = sp.taken(x, '_rick_split')
    return img if (img is not None and _same_tensor(getattr(img, 'scale_of', None), s)) else None


_sup_cache = {}


def _split_geoms_supported(kind, N, I, IH, IW, O, kh):
    """Do the MFMA launches of a modulated layer have split-image forms?  kind 'plain': conv s1 + its dgrad + wgrad;
    'up': convT2 fprop + conv s2 dgrad + wgrad.  -> dict(fprop, dgrad, wgrad)"""
    key = (kind, N, I, IH, IW, O, kh)
    r = _sup_cache.get(key)
    if r is None:
        from .conv import _geom
        p = kh // 2
        if kind == 'plain':
            t3 = [(ky - p, kx - p, ky * kh + kx) for ky in range(kh) for kx in range(kh)]
            t3T = [(p - ky, p - kx, ky * kh + kx) for ky in range(kh) for kx in range(kh)]
            gf = _geom(N, IH, IW, I, IH, IW, O, IH, IW, 1, 1, 0, 0, t3, kh * kh)
            gd = _geom(N, IH, IW, O, IH, IW, I, IH, IW, 1, 1, 0, 0, t3T, kh * kh)
            r = dict(fprop=bool(lib.rick_conv_igemm_split_supported(ctypes.byref(gf))) and I % 32 == 0,
                     dgrad=bool(lib.rick_conv_igemm_split_supported(ctypes.byref(gd))) and O % 32 == 0,
                     wgrad=bool(lib.rick_conv_wgrad_split_supported(ctypes.byref(gf))),
                     nosplitk=lib.rick_conv_igemm_workspace_bytes(ctypes.byref(gf)) == 0)
        else:
            OH, OW = 2 * IH + 1, 2 * IW + 1
            t3s = [(ky, kx, ky * 3 + kx) for ky in range(3) for kx in range(3)]
            gd = _geom(N, OH, OW, O, IH, IW, I, IH, IW, 2, 1, 0, 0, t3s, 9)          # dgrad: stride-2 conv of the output gradient
            r = dict(fprop=kh == 3 and I % 32 == 0 and lib.rick_convt2_workspace_bytes(N, IH, IW, I, O, OH, OW) >= 0,
                     dgrad=kh == 3 and bool(lib.rick_conv_igemm_split_supported(ctypes.byref(gd))) and O % 32 == 0,
                     wgrad=kh == 3 and bool(lib.rick_conv_wgrad_split_supported(ctypes.byref(gd))), nosplitk=True)
        _sup_cache[key] = r
    return r


def demod_coeff(w, s, wscale, eps=1e-8):
    """d[b,o] = rsqrt(sum_{i,k} (wscale*w[o,i,k]*s[b,i])^2 + eps)  (model_probe_tune.py:250).
    Small tensors ([O,I] and [B,I]); plain device tensor algebra, differentiable to any order."""
    wsq = (w * wscale).pow(2).sum([2, 3])          # [O, I]
    return torch.rsqrt(s.pow(2) @ wsq.t() + eps)   # [B, O]


def modulated_conv_composed(x, w, s, d, wscale, upsample, key=None):
    xs = chan_scale(x, s)
    if upsample:
        y = conv_transpose2d(xs, w, stride=2, padding=0, wscale=wscale, key=key)
    else:
        y = conv2d(xs, w, stride=1, padding=w.shape[2] // 2, wscale=wscale, key=key)
    return chan_scale(y, d) if d is not None else y


class _DemodFused(Function):
    """demod_coeff as two launches forward (wsq, d) and two backward (gs; gw) — the tensor-algebra
    form above costs ~8 launches forward and ~16 backward per layer, which is what this latency-bound corner of
    the network is made of.  Under op.second_order() the composed form is used instead."""

    @staticmethod
    def forward(ctx, w, s, wscale, eps, key=None):
        O, I, kh, kw = w.shape
        w_in, s_in = w, s
        w = w.contiguous()
        s = s.contiguous()
        B = s.shape[0]
        wsq = torch.empty((O, I), device=w.device, dtype=w.dtype)
        d = torch.empty((B, O), device=w.device, dtype=w.dtype)
        check(lib.rick_wsq_f32(ptr(w), ptr(wsq), O, I, kh * kw, float(wscale), stream_ptr()), 'rick_wsq_f32')
        check(lib.rick_demod_f32(ptr(s), ptr(wsq), ptr(d), B, I, O, float(eps), stream_ptr()), 'rick_demod_f32')
        ctx.save_for_backward(w, s, wsq, d, w_in, s_in)      # (the inputs themselves: .contiguous() may have copied them)
        ctx.wscale, ctx.eps = float(wscale), float(eps)
        ctx.w_param = param_like(w)
        ctx.key, ctx.sink = key, grad_sink_enabled()
        return d

    @staticmethod
    def backward(ctx, gd):
        w, s, wsq, d, w_in, s_in = ctx.saved_tensors
        O, I, kh, kw = w.shape
        B = s.shape[0]
        if torch.is_grad_enabled():     # create_graph=True: the tensor-algebra form, differentiable to any order
            from ._twice import second_order_backward
            gw, gs = second_order_backward(lambda w_, s_: demod_coeff(w_, s_, ctx.wscale, ctx.eps), (w_in, s_in),
                                           ctx.needs_input_grad[:2], gd, (ctx.w_param, False))
            return gw, gs, None, None, None
        gd = gd.contiguous()
        gw = gs = None
        if ctx.needs_input_grad[1]:
            gs = torch.empty_like(s)
            check(lib.rick_demod_bwd_s_f32(ptr(s), ptr(wsq), ptr(d), ptr(gd), ptr(gs), B, I, O, stream_ptr()),
                  'rick_demod_bwd_s_f32')
        if ctx.needs_input_grad[0]:
            sink = _sink_target(ctx.key, w.shape, ctx.sink)      # op.grad_sink(): add straight into the parameter's .grad
            if sink is not None:
                from .conv import join_side
                join_side()      # (the layer's weight gradient may be accumulating into the same .grad on the side stream)
            gw = torch.empty_like(w) if sink is None else None
            check(lib.rick_demod_bwd_w_f32(ptr(w), ptr(s), ptr(d), ptr(gd), ptr(sink if sink is not None else gw), B, I, O,
                                           kh * kw, ctx.wscale, int(sink is not None), stream_ptr()), 'rick_demod_bwd_w_f32')
        return gw, gs, None, None, None


def demod_coeff_fused(w, s, wscale, eps=1e-8, key=None):
    """Same values as demod_coeff; first-order differentiable; needs B <= 32 (falls back to the composed form).
    `key` = (parameter, tag) lets op.grad_sink() add the weight gradient straight into the parameter's .grad."""
    require_cuda_f32(w, s)
    if s.shape[0] > 32:
        return demod_coeff(w, s, wscale, eps)
    return _DemodFused.apply(w, s, wscale, eps, key)


class _ModConvFused(Function):
    """Fused modulated convolution.  With `tail` = (bias, noise, noise_weight, slope, gain) and a plain (non-upsampling)
    layer, StyledConv's NoiseInjection + bias + LeakyReLU run in the MFMA kernel's epilogue as well, and the output is
    the activation; the backward then starts with the activation adjoint and reconstructs the pre-tail values it needs
    for the demodulation gradient from the saved activation (rick_hw_dot_act_f32)."""

    @staticmethod
    def forward(ctx, x, w, s, d, wscale, upsample, key, bias, noise, nw, slope, gain, next_s=None):
        from . import split as sp
        O, I, kh, kw = w.shape
        s_in, d_in = s, d
        s = s.contiguous()
        d = d.contiguous() if d is not None else None
        tail = bias is not None
        tail_params = (bias, nw)                  # the Parameters themselves (gradient sink targets)
        ctx.plike = (param_like(w), param_like(bias), param_like(nw))
        wp = _pack(w, wscale, key and (key[0], key[1] + ('/convT' if upsample else '/conv')))
        N, _, IH, IW = x.shape
        # Split images (op/split.py): the input's image (its producer folded THIS layer's style in) replaces x and s in every
        # MFMA launch of the layer that has a split form; the output's image (the next layer's style folded in) is written by
        # this layer's own epilogue / its blur.  Exponents come from bounds: |d * conv(s x)| <= sqrt(taps * Ci) * max |x|
        # (Cauchy-Schwarz; the demodulation makes the modulated weight rows unit vectors), then the tail's terms.
        use = _USE_SPLIT and d is not None and s_in is s and kh == kw and x.is_cuda
        sup = _split_geoms_supported('up' if upsample else 'plain', N, I, IH, IW, O, kh) if use else None
        xpk = _input_image(x, s_in) if use else None
        ctx.xpk = xpk if (xpk is not None and (sup['wgrad'] or sup['fprop'])) else None
        ctx.sup = sup
        xin = xpk if (xpk is not None and sup['fprop']) else None
        if upsample:
            if tail:
                raise RuntimeError('the fused tail of an upsampling layer belongs to its blur (upfirdn2d_noise_bias_act)')
            oh, ow = (x.shape[2] - 1) * 2 + kh, (x.shape[3] - 1) * 2 + kw
            if xin is not None:
                A = sp.new_amax(x.device)
                y = _convT_launch(None, wp, O, kh, kw, 2, 0, (oh, ow), oscale=d, x_split=xin, amax=A)
                sp.hand(y, '_rick_bound', (A, 1.0))
                stats['fprop'] += 1
            else:
                y = _convT_launch(x, wp, O, kh, kw, 2, 0, (oh, ow), iscale=s, oscale=d)
                if use:
                    sp.hand(y, '_rick_bound', (_tensor_amax(x), math.sqrt(4.0 * I)))     # (a stride-2 output pixel sees <= 4 of the 9 taps)
        elif tail:
            bias, noise, nw = bias.contiguous(), noise.contiguous(), nw.contiguous()
            epi = _epilogue(bias, noise, nw, slope, gain)
            if use:
                A = sp.new_amax(x.device)
                epi.amax = ptr(A)
                img = None
                if next_s is not None and sup['nosplitk'] and O % 4 == 0:
                    ns = next_s.contiguous()
                    bound = _bound_tail(_tensor_amax(x), math.sqrt(kh * kw * I), gain, nw, _tensor_amax(noise), bias, ns)
                    img = sp.SplitImage(torch.empty((N, O, IH, IW), device=x.device, dtype=torch.float32,
                                                    memory_format=torch.channels_last), sp.new_words(4, x.device), (bound, None, 1.0))
                    img.scale_of = next_s
                    epi.split_out, epi.split_hdr, epi.split_bound, epi.split_coef, epi.split_scale = (
                        ptr(img.data), ptr(img.hdr), ptr(bound), 1.0, ptr(ns))
            if xin is not None:
                stats['fprop'] += 1
                y = _conv_launch(None, wp, O, kh, kw, 1, kh // 2, oscale=d, epi=epi, x_split=xin)
            else:
                y = _conv_launch(x, wp, O, kh, kw, 1, kh // 2, iscale=s, oscale=d, epi=epi)
            if use:
                sp.hand(y, '_rick_amax', A)
                if img is not None:
                    sp.hand(y, '_rick_split', img)
                    stats['produced'] += 1
        else:
            y = _conv_launch(x, wp, O, kh, kw, 1, kh // 2, iscale=s, oscale=d)
        ctx.save_for_backward(x, w, s, d, y, *((bias, noise, nw) if tail else ()))
        ctx.sd_in = None if (s is s_in and d is d_in) else (s_in, d_in)     # only when .contiguous() copied (never in the networks)
        ctx.tail_params = tail_params
        ctx.cfg = (wscale, upsample, key, tail, slope, gain)
        ctx.sink = grad_sink_enabled()
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, s, d, y = ctx.saved_tensors[:5]
        wscale, upsample, key, tail, slope, gain = ctx.cfg
        O, I, kh, kw = w.shape
        if torch.is_grad_enabled():     # create_graph=True (path length): differentiate the composed form
            from ._twice import second_order_backward
            from .fused_act import fused_noise_bias_act
            bias, nw = ctx.tail_params
            noise = ctx.saved_tensors[6] if tail else None
            if ctx.sd_in is not None:
                s, d = ctx.sd_in

            def compose(x_, w_, s_, d_, b_, nw_):
                yc = modulated_conv_composed(x_, w_, s_, d_, wscale, upsample, key)
                return fused_noise_bias_act(yc, b_, noise, nw_, slope, gain) if tail else yc
            gx, gw, gs, gd, gb, gnw = second_order_backward(
                compose, (x, w, s, d, bias, nw), [ctx.needs_input_grad[i] for i in (0, 1, 2, 3, 7, 9)], g,
                (False, ctx.plike[0], False, False, ctx.plike[1], ctx.plike[2]))
            return (gx, gw, gs, gd, None, None, None, gb, None, gnw, None, None, None)
        g_in = g
        g = g.contiguous(memory_format=torch.channels_last)
        gx = gs = gw = gd = gb = gnw = None
        sup, xpk = ctx.sup, ctx.xpk
        gpk = None                                      # split image of d * (gradient of the convolution's output)
        if tail:
            from .fused_act import _ActAdjoint, param_sink
            bias, noise, nw = ctx.saved_tensors[5:]
            want_b, want_w = ctx.needs_input_grad[7], ctx.needs_input_grad[9]
            sink_b = param_sink(ctx.tail_params[0], O, ctx.sink and want_b)
            sink_w = param_sink(ctx.tail_params[1], 1, ctx.sink and want_w)
            gd_fused = None
            if sup is not None and sup['dgrad']:
                # the adjoint leaves as fp32 (the demodulation gradient reads it) AND as the image of d * adjoint
                g, gpk, gb, gnw = _adjoint_with_image(g, y, noise, slope, gain, d, want_b, want_w, sink_b, sink_w)
            else:
                res = None
                if _FUSE_ADJOINT_DOT and d is not None and ctx.needs_input_grad[3]:
                    # ... and the demodulation gradient from the same pass over (g, y) instead of a second one
                    from .fused_act import act_adjoint_dot
                    res = act_adjoint_dot(g, y, noise, slope, gain, want_b, want_w, sink_b, sink_w, bias, nw, d)
                if res is not None:
                    g, gb, gnw, gd_fused = res
                else:
                    g, gb, gnw = _ActAdjoint.apply(g, y, noise, slope, gain, want_b, want_w, sink_b, sink_w)
        elif sup is not None and sup['dgrad']:
            from . import split as sp
            img = sp.taken(g_in, '_rick_split')    # written by the blur's adjoint (op/upfirdn2d.py) with THIS d folded in
            if img is not None and _same_tensor(getattr(img, 'scale_of', None), d):
                gpk = img
        wT = w.transpose(0, 1)
        wpT = _pack(wT, wscale, key and (key[0], key[1] + '/T'))
        stats['dgrad'] += gpk is not None
        # unscaled data gradient: gx' = W^T (g * d)
        if upsample:
            gxu = _conv_launch(None, wpT, I, kh, kw, 2, 0, x_split=gpk) if gpk is not None else _conv_launch(g, wpT, I, kh, kw, 2, 0, iscale=d)
        elif gpk is not None:
            gxu = _convT_launch(None, wpT, I, kh, kw, 1, kh // 2, (x.shape[2], x.shape[3]), x_split=gpk)
        else:
            gxu = _convT_launch(g, wpT, I, kh, kw, 1, kh // 2, (x.shape[2], x.shape[3]), iscale=d)
        if ctx.needs_input_grad[2] and ctx.needs_input_grad[0] and I % 4 == 0:
            gs, gx = _hw_dot_scale_raw(gxu, x, s)          # one pass over gxu: style gradient + scaled data gradient
        else:
            if ctx.needs_input_grad[2]:
                gs = _hw_dot_raw(gxu, x)
            if ctx.needs_input_grad[0]:
                gx = _chan_scale_raw(gxu, s)
        if ctx.needs_input_grad[1]:
            sink = _sink_target(key, w.shape, ctx.sink)     # op.grad_sink(): add straight into the parameter's .grad
            both = gpk is not None and xpk is not None and sup['wgrad']       # both operands as images (scales folded in)
            stats['wgrad'] += both
            if upsample:   # convT: gw[o,i,k] = sum x[pos,i] g[pos*2+k, o]  (a = x, b = g), transposed back
                if both:
                    gw = _wgrad_launch(None, None, kh, kw, 2, 0, wscale, out=sink, transposed=True, a_split=xpk, b_split=gpk)
                else:
                    gw = _wgrad_launch(x, g, kh, kw, 2, 0, wscale, ascale=s, bscale=d, out=sink, transposed=True)
                gw = gw.transpose(0, 1) if gw is not None else None
            elif both:
                gw = _wgrad_launch(None, None, kh, kw, 1, kh // 2, wscale, out=sink, a_split=gpk, b_split=xpk)
            else:
                gw = _wgrad_launch(g, x, kh, kw, 1, kh // 2, wscale, ascale=d, bscale=s, out=sink)
        if d is not None and ctx.needs_input_grad[3]:
            # conv_out = d * y'  ->  sum g*y' = (sum g*conv_out) / d,  d > 0
            if tail and gd_fused is not None:
                gd = gd_fused
            elif tail:
                gd = _hw_dot_act_raw(g, y, bias, noise, nw, slope, gain, divisor=d)
            else:
                gd = _hw_dot_raw(g, y, divisor=d)
        if gx is not None:
            gx._rick_owned = True        # a fresh buffer nobody else holds: op.torgb_fork may accumulate into it in place
        return (gx, gw, gs, gd, None, None, None, gb, None, gnw, None, None, None)


def _hw_dot_scale_raw(a, b, scale):
    """(sum_hw a*b [N,C], a * scale[n,c]) from one read of a."""
    a = a.contiguous(memory_format=torch.channels_last)
    b = b.contiguous(memory_format=torch.channels_last)
    n, c, h, w = a.shape
    out = torch.empty((n, c), device=a.device, dtype=a.dtype)
    scaled = torch.empty_like(a)
    part = torch.empty(lib.rick_hw_dot_blocks(h * w) * n * c, device=a.device, dtype=a.dtype)
    check(lib.rick_hw_dot_scale_f32(ptr(a), ptr(b), ptr(out), ptr(scale), ptr(scaled), n, h * w, c, ptr(part), stream_ptr()),
          'rick_hw_dot_scale_f32')
    return out, scaled


def _hw_dot_act_raw(g, y, bias, noise, nw, slope, gain, divisor=None):
    n, c, h, w = g.shape
    out = torch.empty((n, c), device=g.device, dtype=g.dtype)
    part = torch.empty(lib.rick_hw_dot_blocks(h * w) * n * c, device=g.device, dtype=g.dtype)
    check(lib.rick_hw_dot_act_f32(ptr(g), ptr(y), ptr(out), n, h * w, c, ptr(bias), ptr(noise), ptr(nw), noise.shape[0],
                                  float(slope), float(gain), ptr(part), ptr(divisor), stream_ptr()), 'rick_hw_dot_act_f32')
    return out


def _adjoint_with_image(g, y, noise, slope, gain, d, want_b, want_w, sink_b, sink_w):
    """The activation adjoint gz = g * act'(y) * gain as fp32 AND as the split image of d[n,c] * gz (the operand of the
    layer's data- and weight-gradient kernels), bias / noise-strength gradients as in _ActAdjoint (sunk into the parameters'
    .grad when sinks are given).  -> (gz, image, gb, gnw)"""
    from . import split as sp
    from .fused_act import _noise_args
    n, c, h, w = g.shape
    rows = n * h * w
    Ag = _tensor_amax(g)
    bound = _bound_tail(Ag, 1.0, abs(gain) * max(1.0, abs(slope)), mul=d)
    gz = torch.empty_like(g)
    img = sp.SplitImage(torch.empty_like(g), sp.new_words(4, g.device), (bound, None, 1.0))
    img.scale_of = d
    want_w = want_w and noise is not None
    nz, nb, nhw = _noise_args(noise, g) if want_w else (None, 1, 1)
    if (want_b and sink_b is None) or (want_w and sink_w is None):
        sink_b = sink_w = None                         # one accumulate switch serves both sums (as in _ActAdjoint)
    sunk = (want_b and sink_b is not None) or (want_w and sink_w is not None)
    gb = (sink_b if sunk else torch.empty(c, device=g.device, dtype=g.dtype)) if want_b else None
    gw = (sink_w if sunk else torch.empty(1, device=g.device, dtype=g.dtype)) if want_w else None
    part = None
    if want_b or want_w:
        part = torch.empty(lib.rick_bias_act_bwd_blocks(rows, c) * (c + 1), device=g.device, dtype=g.dtype)
    check(lib.rick_bias_act_bwd_split2_f32(ptr(g), ptr(y), ptr(img.data), ptr(img.hdr), None, None, 0.0, None, ptr(d), ptr(bound),
                                           ptr(gz), ptr(gb), ptr(gw), ptr(nz), rows, c, h * w, nb, nhw, float(slope), float(gain),
                                           ptr(part), int(sunk), stream_ptr()), 'rick_bias_act_bwd_split2_f32')
    return gz, img, (None if sunk else gb), (None if sunk else gw)


def modulated_conv_fused(x, w, s, d, wscale, upsample, key=None, tail=None, next_s=None):
    """tail = (bias, noise, noise_weight, negative_slope, gain) fuses StyledConv's activation tail (plain layers only).
    next_s: the style scales [N, Co] of the NEXT modulated convolution — the layer then also writes its output as that
    convolution's split-image operand (first-order steps)."""
    if tail is None:
        return _ModConvFused.apply(x, w, s, d, float(wscale), bool(upsample), key, None, None, None, 0.2, 1.0, None)
    bias, noise, nw, slope, gain = tail
    return _ModConvFused.apply(x, w, s, d, float(wscale), bool(upsample), key, bias, noise, nw, float(slope), float(gain), next_s)


# ------------------------------------------------------------------------------------ modulation bank
# rick_modbank_desc (include/rick_hip.h), 56 bytes
_MB_DESC = np.dtype([('w', '<u8'), ('b', '<u8'), ('io_off', '<i8'), ('gw_off', '<i8'), ('gb_off', '<i8'), ('C', '<i4'),
                     ('lat_idx', '<i4'), ('blk_begin', '<i4'), ('reserved', '<i4')])


class ModulationBank:
    """The style -> per-channel scale linears of every modulated convolution of a generator
    (``ModulatedConv2d.modulation``, model_probe_tune.py:233,246) evaluated by ONE launch, their weight / bias gradients
    by one more (rick_modbank_{fwd,bwd}_f32) — the per-layer form costs a rocBLAS GEMM forward and two GEMMs, a column
    sum and several scalings backward, ~180 launches of 4-8 us per generator forward + backward.  `linears` are the
    EqualLinear modules in layer order, `lat_idx` the latent row each of them reads."""

    def __init__(self, linears, lat_idx):
        self.linears, self.lat_idx = list(linears), list(lat_idx)
        self.C = [m.weight.shape[0] for m in self.linears]
        self.K = self.linears[0].weight.shape[1]
        self.scale = self.linears[0].scale
        if any(m.weight.shape[1] != self.K or m.scale != self.scale or m.lr_mul != 1 or m.activation for m in self.linears):
            raise RuntimeError('ModulationBank: layers must share style_dim / scale and have no activation')
        self._tables = {}      # batch -> (device table, pointer signature); old tables are kept (graphs may read them)
        self._retired = []
        # gradient layout: [W_0 | b_0 | W_1 | b_1 ...], each start 16-byte aligned
        self.gw_off, self.gb_off, pos = [], [], 0
        for c in self.C:
            self.gw_off.append(pos)
            pos += c * self.K
            self.gb_off.append(pos)
            pos += (c + 3) // 4 * 4
        self.grad_floats = pos
        self.blk_begin, blk = [], 0
        for c in self.C:
            self.blk_begin.append(blk)
            blk += lib.rick_modbank_blocks(c)
        self.total_blocks = blk

    def params(self):
        out = []
        for m in self.linears:
            out += [m.weight, m.bias]
        return out

    def table(self, B, device):
        sig = tuple(p.data_ptr() for p in self.params())
        ent = self._tables.get(B)
        if ent is None or ent[1] != sig:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('ModulationBank: descriptor table must be built before hipGraph capture (run the step eagerly once)')
            arr = np.zeros(len(self.linears), dtype=_MB_DESC)
            io = 0
            for i, m in enumerate(self.linears):
                arr[i] = (m.weight.data_ptr(), m.bias.data_ptr(), io, self.gw_off[i], self.gb_off[i], self.C[i], self.lat_idx[i],
                          self.blk_begin[i], 0)
                io += B * self.C[i]
            if ent is not None:
                self._retired.append(ent[0])
            ent = (torch.from_numpy(arr.view(np.uint8).copy()).to(device), sig)
            self._tables[B] = ent
        return ent[0]

    def sink_table(self, B, device):
        """(device table, base address) whose gradient offsets address the parameters' OWN ``.grad`` buffers relative to
        `base` (op.grad_sink(): the backward launch adds into them), -1 for parameters that take no gradient; None when
        some parameter that needs a gradient has no suitable buffer."""
        need = [p.requires_grad for p in self.params()]
        grads = [p.grad if n else None for p, n in zip(self.params(), need)]
        if any(n and (g is None or not p.is_leaf or not g.is_contiguous() or g.data_ptr() % 16 or g.dtype != torch.float32)
               for p, g, n in zip(self.params(), grads, need)) or not any(need):
            return None
        sig = (tuple(p.data_ptr() for p in self.params()), tuple(g.data_ptr() if g is not None else 0 for g in grads))
        ent = self._tables.get(('sink', B))
        if ent is None or ent[1] != sig:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('ModulationBank: descriptor table must be built before hipGraph capture (run the step eagerly once)')
            base = min(g.data_ptr() for g in grads if g is not None)
            arr = np.zeros(len(self.linears), dtype=_MB_DESC)
            io = 0
            for i, m in enumerate(self.linears):
                gw, gb = grads[2 * i], grads[2 * i + 1]
                arr[i] = (m.weight.data_ptr(), m.bias.data_ptr(), io, (gw.data_ptr() - base) // 4 if gw is not None else -1,
                          (gb.data_ptr() - base) // 4 if gb is not None else -1, self.C[i], self.lat_idx[i], self.blk_begin[i], 0)
                io += B * self.C[i]
            if ent is not None:
                self._retired.append(ent[0])
            ent = (torch.from_numpy(arr.view(np.uint8).copy()).to(device), sig, base)
            self._tables[('sink', B)] = ent
        return ent[0], ent[2]

    def __call__(self, latent):
        """latent [B, n_latent, K] (no gradient is produced for it) -> list of s_l [B, C_l]."""
        require_cuda_f32(latent)
        return list(_ModBank.apply(latent.contiguous(), self, *self.params()))


class _ModBank(Function):
    @staticmethod
    def forward(ctx, latent, bank, *params):
        B, n_latent, K = latent.shape
        if K != bank.K or B > 8:
            raise RuntimeError('ModulationBank: unsupported latent shape')
        tab = bank.table(B, latent.device)
        out = torch.empty(B * sum(bank.C), device=latent.device, dtype=latent.dtype)
        check(lib.rick_modbank_fwd_f32(ptr(latent), B, n_latent, K, ptr(tab), len(bank.C), bank.total_blocks, bank.scale, ptr(out),
                                       stream_ptr()), 'rick_modbank_fwd_f32')
        ctx.save_for_backward(latent)
        ctx.bank = bank
        ctx.sink = grad_sink_enabled()
        ctx.set_materialize_grads(False)
        res, off, dead = [], 0, []
        for i, c in enumerate(bank.C):
            res.append(out[off:off + B * c].view(B, c))
            off += B * c
            # a layer whose modulation parameters are frozen (ToRGB: the optimiser owns only `convs.*`,
            # train_dynamic_update_prune.py:908-917) must not make its consumer compute a style gradient
            if not (params[2 * i].requires_grad or params[2 * i + 1].requires_grad):
                dead.append(res[-1])
        if dead:
            ctx.mark_non_differentiable(*dead)
        return tuple(res)

    @staticmethod
    def backward(ctx, *gs):
        (latent,) = ctx.saved_tensors
        bank = ctx.bank
        B, n_latent, K = latent.shape
        if torch.is_grad_enabled():     # create_graph=True: per-layer linears (plain tensor algebra)
            from ._twice import second_order_backward
            params = bank.params()

            def compose(lat_, _bank, *ps):
                return tuple(torch.addmm(ps[2 * i + 1], lat_[:, bank.lat_idx[i]], ps[2 * i].t(), alpha=bank.scale)
                             for i in range(len(bank.C)))
            res = second_order_backward(compose, [latent, None] + params, ctx.needs_input_grad, gs)
            return tuple(res)
        if ctx.needs_input_grad[0]:
            raise RuntimeError('ModulationBank produces no latent gradient; use the per-layer path when the latent requires grad')
        need = ctx.needs_input_grad[2:]
        live = [need[2 * i] or need[2 * i + 1] for i in range(len(bank.C))]
        if not any(live):
            return (None,) * (2 + len(need))
        if any(l and g is None for l, g in zip(live, gs)):       # an unused output of a trainable layer: zero gradient
            gs = [g if (g is not None or not l) else latent.new_zeros(B, c) for g, l, c in zip(gs, live, bank.C)]
        # frozen layers (their slice of `flat` is never read) get an uninitialised placeholder: no fill launch
        flat = torch.cat([(g if l else latent.new_empty(B, c)).reshape(-1) for g, l, c in zip(gs, live, bank.C)])
        sink = bank.sink_table(B, latent.device) if ctx.sink else None
        if sink is not None:
            # op.grad_sink(): the launch adds every weight / bias gradient straight into the parameter's .grad
            check(lib.rick_modbank_bwd_f32(ptr(latent), ptr(flat), B, n_latent, K, ptr(sink[0]), len(bank.C), bank.total_blocks,
                                           bank.scale, sink[1], 1, stream_ptr()), 'rick_modbank_bwd_f32')
            return (None,) * (2 + len(need))
        grad = torch.empty(bank.grad_floats, device=latent.device, dtype=latent.dtype)
        check(lib.rick_modbank_bwd_f32(ptr(latent), ptr(flat), B, n_latent, K, ptr(bank.table(B, latent.device)), len(bank.C),
                                       bank.total_blocks, bank.scale, ptr(grad), 0, stream_ptr()), 'rick_modbank_bwd_f32')
        res = [None, None]
        for i, c in enumerate(bank.C):
            need_w, need_b = ctx.needs_input_grad[2 + 2 * i], ctx.needs_input_grad[3 + 2 * i]
            res.append(grad[bank.gw_off[i]:bank.gw_off[i] + c * K].view(c, K) if need_w else None)
            res.append(grad[bank.gb_off[i]:bank.gb_off[i] + c] if need_b else None)
        return tuple(res)


# ------------------------------------------------------------------------------------------------------------------
# rick_demod_desc (include/rick_hip.h), 72 bytes
_DM_DESC = np.dtype([('w', '<u8'), ('wsq', '<u8'), ('gw', '<u8'), ('s_off', '<i8'), ('d_off', '<i8'), ('O', '<i4'), ('I', '<i4'),
                     ('K', '<i4'), ('scale2', '<f4'), ('blk_wsq', '<i4'), ('blk_demod', '<i4'), ('blk_bwd_s', '<i4'),
                     ('reserved', '<i4')])


class DemodBank:
    """The demodulation coefficients d_l = rsqrt(sum_i s_l^2 wsq_l + eps) of every demodulated convolution of a generator
    (model_probe_tune.py:246-252) from ONE launch, their style and weight gradients from one launch each, and
    wsq_l = scale^2 sum_k w_l^2 — a function of the weights only — recomputed once per update of the network instead of
    in every forward (behind the weight packs, PackGroup.after_repack).  Per layer the four kernels of `_DemodFused` cost
    13 x (5.6 + 6.0) us per forward and 13 x (9.6 + 10.2) us per backward at 256 px: 0.43 ms of a train iteration.  Same
    arithmetic and summation order per element: the coefficients and gradients are bit-identical to the per-layer path.

    `convs`: the ModulatedConv2d modules with demodulate=True in forward order; `s_index[l]`: position of layer l's style
    vector in the ModulationBank's output list; `mod_C`: channel count of every entry of that list (its flat layout)."""

    def __init__(self, convs, s_index, mod_C):
        self.convs, self.s_index, self.mod_C = list(convs), list(s_index), list(mod_C)
        self.O = [c.weight.shape[1] for c in self.convs]
        self.I = [c.weight.shape[2] for c in self.convs]
        self.K = [c.weight.shape[3] * c.weight.shape[4] for c in self.convs]
        self.eps = self.convs[0].eps
        if any(c.eps != self.eps or not c.demodulate or self.mod_C[j] != i for c, j, i in zip(self.convs, self.s_index, self.I)):
            raise RuntimeError('DemodBank: layers must share eps and read a style vector of their input width')
        self.blk_wsq, self.blk_demod, self.blk_bwd_s = [], [], []
        a = b = c = 0
        for o, i in zip(self.O, self.I):
            self.blk_wsq.append(a)
            self.blk_demod.append(b)
            self.blk_bwd_s.append(c)
            a += lib.rick_demod_blocks_wsq(o, i)
            b += lib.rick_demod_blocks(o)
            c += lib.rick_demod_blocks_bwd_s(i)
        self.total_wsq, self.total_demod, self.total_bwd_s = a, b, c
        self._wsq = None            # persistent [O_l * I_l] buffers
        self._stamps = None
        self._tables = {}           # (B, sink) -> (device table, signature)
        self._retired = []
        self._hooked = None

    def weights(self):
        return [c.weight for c in self.convs]

    def offsets(self, B):
        mod_off, pos = [], 0
        for c in self.mod_C:
            mod_off.append(pos)
            pos += B * c
        d_off, q = [], 0
        for o in self.O:
            d_off.append(q)
            q += B * o
        return [mod_off[j] for j in self.s_index], d_off, pos, q

    def _sinks(self):
        out = []
        for w in self.weights():
            g = w.grad
            ok = (w.is_leaf and w.requires_grad and g is not None and g.is_contiguous() and g.dtype == torch.float32
                  and g.numel() == w.numel())
            out.append(g if ok else None)
        return out

    def table(self, B, device, sink=False):
        ws = self.weights()
        if self._wsq is None or self._wsq[0].device != device:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('DemodBank: buffers must exist before hipGraph capture (run the step eagerly once)')
            self._wsq = [torch.empty(o * i, device=device, dtype=torch.float32) for o, i in zip(self.O, self.I)]
            self._stamps = None
        grads = self._sinks() if sink else [None] * len(ws)
        sig = (tuple(w.data_ptr() for w in ws), tuple(g.data_ptr() if g is not None else 0 for g in grads),
               tuple(q.data_ptr() for q in self._wsq))
        ent = self._tables.get((B, sink))
        if ent is None or ent[1] != sig:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('DemodBank: descriptor table must be built before hipGraph capture (run the step eagerly once)')
            s_off, d_off, _, _ = self.offsets(B)
            arr = np.zeros(len(ws), dtype=_DM_DESC)
            for l, (w, c) in enumerate(zip(ws, self.convs)):
                arr[l] = (w.data_ptr(), self._wsq[l].data_ptr(), grads[l].data_ptr() if grads[l] is not None else 0, s_off[l],
                          d_off[l], self.O[l], self.I[l], self.K[l], np.float32(c.scale) * np.float32(c.scale), self.blk_wsq[l],
                          self.blk_demod[l], self.blk_bwd_s[l], 0)
            if ent is not None:
                self._retired.append(ent[0])
            ent = (torch.from_numpy(arr.view(np.uint8).copy()).to(device), sig)
            self._tables[(B, sink)] = ent
        return ent[0]

    def _current(self):
        from .conv import weights_stamp
        return [weights_stamp(w) for w in self.weights()]

    def refresh_wsq(self, force=False):
        """wsq of every layer in one launch if any weight changed since the last one (host side; PackGroup.after_repack calls
        it whenever the network is repacked, __call__ when it finds the stamps stale)."""
        ws = self.weights()
        if not ws[0].is_cuda:
            return
        cur = self._current()
        if not force and self._stamps == cur and self._wsq is not None:
            return
        tab = self.table(1, ws[0].device)
        check(lib.rick_wsq_multi_f32(ptr(tab), len(ws), self.total_wsq, stream_ptr()), 'rick_wsq_multi_f32')
        self._stamps = cur

    def _hook(self):
        grp = getattr(self.weights()[0], '_rick_group', None)
        if grp is not None and self._hooked is not grp:
            import weakref
            ref = weakref.ref(self)
            grp.after_repack.append(lambda: ref() is not None and ref().refresh_wsq())
            self._hooked = grp

    def usable(self, s_list):
        """The style vectors must be the ModulationBank's own views (one flat buffer in its layout), contiguous float32."""
        B = s_list[self.s_index[0]].shape[0]
        if B > 8 or any(not w.is_contiguous() for w in self.weights()):
            return False
        s_off, _, _, _ = self.offsets(B)
        s0 = s_list[self.s_index[0]]
        base = s0.data_ptr() - 4 * s_off[0]
        return all(s_list[j].dtype == torch.float32 and s_list[j].is_contiguous() and s_list[j].data_ptr() == base + 4 * o
                   for j, o in zip(self.s_index, s_off))

    def __call__(self, s_list):
        """s_list: the ModulationBank's outputs -> list of d_l [B, O_l] (None when the inputs do not have the bank's layout)."""
        if not self.usable(s_list):
            return None
        self._hook()
        self.refresh_wsq()
        return list(_DemodBankFn.apply(self, *[s_list[j] for j in self.s_index], *self.weights()))


class _DemodBankFn(Function):
    @staticmethod
    def forward(ctx, bank, *args):
        L = len(bank.convs)
        ss, ws = args[:L], args[L:]
        B = ss[0].shape[0]
        dev = ss[0].device
        s_off, d_off, _, d_total = bank.offsets(B)
        tab = bank.table(B, dev)
        d_flat = torch.empty(d_total, device=dev, dtype=torch.float32)
        s_base = ss[0].data_ptr() - 4 * s_off[0]
        check(lib.rick_demod_multi_f32(s_base, ptr(d_flat), ptr(tab), L, bank.total_demod, B, max(bank.I), float(bank.eps),
                                       stream_ptr()), 'rick_demod_multi_f32')
        ctx.bank, ctx.B = bank, B
        ctx.sink = grad_sink_enabled()
        ctx.w_param = [param_like(w) for w in ws]
        ctx.save_for_backward(d_flat, *ss, *ws)
        ctx.set_materialize_grads(False)
        return tuple(d_flat[o:o + B * n].view(B, n) for o, n in zip(d_off, bank.O))

    @staticmethod
    def backward(ctx, *gds):
        bank, B = ctx.bank, ctx.B
        L = len(bank.convs)
        d_flat, *rest = ctx.saved_tensors
        ss, ws = rest[:L], rest[L:]
        need_s, need_w = ctx.needs_input_grad[1:1 + L], ctx.needs_input_grad[1 + L:]
        if torch.is_grad_enabled():     # create_graph=True: the tensor-algebra form per layer, differentiable to any order
            from ._twice import second_order_backward

            def compose(*a):
                return tuple(demod_coeff(a[L + l].view(bank.O[l], bank.I[l], *bank.convs[l].weight.shape[3:]), a[l],
                                         bank.convs[l].scale, bank.eps) for l in range(L))
            res = second_order_backward(compose, list(ss) + list(ws), list(need_s) + list(need_w), gds,
                                        tuple([False] * L + list(ctx.w_param)))
            return (None, *res)
        if not (any(need_s) or any(need_w)) or all(g is None for g in gds):
            return (None,) * (1 + 2 * L)
        dev = d_flat.device
        s_off, d_off, s_total, _ = bank.offsets(B)
        gd_flat = torch.cat([(g.contiguous() if g is not None else d_flat.new_zeros(B * o)).reshape(-1) for g, o in zip(gds, bank.O)])
        s_base = ss[0].data_ptr() - 4 * s_off[0]
        gs = [None] * L
        if any(need_s):
            gs_flat = torch.empty(s_total, device=dev, dtype=torch.float32)
            check(lib.rick_demod_bwd_s_multi_f32(s_base, ptr(d_flat), ptr(gd_flat), ptr(gs_flat), ptr(bank.table(B, dev)), L,
                                                 bank.total_bwd_s, B, max(bank.O), stream_ptr()), 'rick_demod_bwd_s_multi_f32')
            gs = [gs_flat[o:o + B * i].view(B, i) if n else None for o, i, n in zip(s_off, bank.I, need_s)]
        gw = [None] * L
        if any(need_w):
            sinks = bank._sinks() if ctx.sink else [None] * L
            if all(g is not None for g, n in zip(sinks, need_w) if n):
                # op.grad_sink(): one launch adds every layer's gradient straight into the parameter's .grad (frozen layers and
                # layers that need none carry a NULL pointer in the table and are skipped)
                if any(g is not None and not n for g, n in zip(sinks, need_w)):
                    sinks = None       # (a .grad buffer on a weight that must not receive a gradient: per-layer path below)
            else:
                sinks = None
            if sinks is not None:
                from .conv import join_side
                join_side()      # the layers' own weight gradients (side stream, op.wgrad_overlap) write the same .grad buffers
                check(lib.rick_demod_bwd_w_multi_f32(s_base, ptr(d_flat), ptr(gd_flat), ptr(bank.table(B, dev, sink=True)), L,
                                                     bank.total_wsq, B, stream_ptr()), 'rick_demod_bwd_w_multi_f32')
            else:
                for l in range(L):
                    if not need_w[l]:
                        continue
                    w = ws[l]
                    gw[l] = torch.empty_like(w)
                    dl = d_flat[d_off[l]:d_off[l] + B * bank.O[l]]
                    gl = gd_flat[d_off[l]:d_off[l] + B * bank.O[l]]
                    check(lib.rick_demod_bwd_w_f32(ptr(w), ptr(ss[l]), ptr(dl), ptr(gl), ptr(gw[l]), B, bank.I[l], bank.O[l],
                                                   bank.K[l], float(bank.convs[l].scale), 0, stream_ptr()), 'rick_demod_bwd_w_f32')
        return (None, *gs, *gw)
