"""The discriminator ResBlock (model_probe_tune.py:644-660) as ONE autograd node on split images.

    t1  = lrelu(conv3x3(x) + b1) * sqrt2                       ConvLayer(in, in, 3)
    t2  = lrelu(conv3x3_s2(blur(t1)) + b2) * sqrt2             ConvLayer(in, out, 3, downsample=True)
    sk  = conv1x1(blur_down2(x))                               ConvLayer(in, out, 1, downsample=True, no act / bias)
    out = (t2 + sk) / sqrt2

Same kernels and values as the per-layer path (rick_amd/models.py: conv2d_bias_act, upfirdn2d, conv2d, add_scale); what
changes is what travels between them (rick_amd/op/split.py): every tensor whose only consumers are MFMA kernels — the
blurred maps, the activation adjoints — exists only as a split image written by its producer, the block's input and output
carry one next to the fp32 tensor, and no convolution kernel of the block converts fp32 operands (2/3 of the conv work of
an iteration is the discriminator's).  The exponents come from exact running maxima measured by the producing kernels'
epilogues (atomic max) and combined by the triangle inequality — never from samples.  The backward adds the skip path's
data gradient into the main path's inside the FIR launch (no autograd add pass) and hands its own maximum to the next
block through the gradient tensor.  First order only: with grad mode enabled in backward (create_graph=True) the block
differentiates the per-layer, twice-differentiable composition instead (op/_twice.py); `op.second_order()` callers never
get here."""
import ctypes
import math

import torch
from torch.autograd import Function

from .._lib import SplitOut, check, lib, ptr, stream_ptr
from . import split as sp
from .conv import (_conv_launch, _convT_launch, _epilogue, _geom, _pack, _sink_target, _wgrad_launch, conv_out_size,
                   grad_sink_enabled, param_like, skip_param_grad)
from .fused_act import param_sink
from .upfirdn2d import _flipped

_SQ = 1.0 / math.sqrt(2.0)


def _fir_ex(x, taps, up, down, pad4, out=None, split_bound=None, bound1=None, coef=1.0, amax=None, accumulate=False, no_f32=False,
            chan_scale=None):
    """upfirdn2d (channels-last) with the extended result handling -> (fp32 tensor or None, SplitImage or None)."""
    n, c, h, w = x.shape
    kh, kw = taps.shape
    oh = (h * up + pad4[2] + pad4[3] - kh) // down + 1
    ow = (w * up + pad4[0] + pad4[1] - kw) // down + 1
    ex = SplitOut()
    img = None
    if split_bound is not None:
        data = torch.empty((n, c, oh, ow), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        img = sp.SplitImage(data, sp.new_words(4, x.device), (split_bound, bound1, float(coef)))
        ex.split_out, ex.split_hdr, ex.bound0, ex.bound1, ex.bound_coef = ptr(data), ptr(img.hdr), ptr(split_bound), ptr(bound1), float(coef)
    ex.amax, ex.accumulate, ex.no_f32, ex.chan_scale = ptr(amax), int(accumulate), int(no_f32), ptr(chan_scale)
    if out is None and not no_f32:
        out = torch.empty((n, c, oh, ow), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
    from .conv import hbm_launch
    nbytes = 4 * (x.numel() + n * c * oh * ow * ((0 if no_f32 else 1) + (1 if accumulate else 0) + (1 if img is not None else 0)))
    check(hbm_launch('upfirdn2d', nbytes, lib.rick_upfirdn2d_ex_f32, ptr(x), ptr(taps), ptr(out), n, h, w, c, kh, kw, up, up, down, down,
                     pad4[0], pad4[1], pad4[2], pad4[3], None, ctypes.byref(ex), stream_ptr()), 'rick_upfirdn2d_ex_f32')
    return out, img


def _act_adjoint_split(g, y, slope, scale, amax_g, mul2=None, want_b=False, sink_b=None):
    """The activation adjoint as split images: (image of g * act'(y) * scale, image of g * mul2 or None, gb or None)."""
    n, c, h, w = g.shape
    rows = n * h * w
    out1 = sp.SplitImage(torch.empty_like(g), sp.new_words(4, g.device), (amax_g, None, abs(scale) * max(1.0, abs(slope))))
    out2 = sp.SplitImage(torch.empty_like(g), sp.new_words(4, g.device), (amax_g, None, abs(mul2))) if mul2 is not None else None
    gb = part = None
    sunk = want_b and sink_b is not None
    if want_b:
        gb = sink_b if sunk else torch.empty(c, device=g.device, dtype=g.dtype)
        part = torch.empty(lib.rick_bias_act_bwd_blocks(rows, c) * (c + 1), device=g.device, dtype=g.dtype)
    from .conv import hbm_launch
    from .fused_act import defer_act_sums
    later = sunk and defer_act_sums(part, lib.rick_bias_act_bwd_blocks(rows, c), c, gb, None)
    check(hbm_launch('bias_act_bwd', 4 * g.numel() * (3 if out2 is None else 4), lib.rick_bias_act_bwd_split_f32,
                     ptr(g), ptr(y), ptr(out1.data), ptr(out1.hdr), ptr(out2.data) if out2 else None,
                     ptr(out2.hdr) if out2 else None, float(mul2 or 0.0), ptr(amax_g), ptr(gb), None, None,
                     rows, c, h * w, 1, 1, float(slope), float(scale), ptr(part), int(sunk) | (2 if later else 0), stream_ptr()),
          'rick_bias_act_bwd_split_f32')
    return out1, out2, (None if sunk else gb)


def _fir_adjoint_split(g, taps, pad4, y, slope, gain, amax_g, want_b=False, sink_b=None):
    """(blur adjoint -> activation adjoint) in ONE launch: the 4x4 FIR of g (flipped taps, pads of the adjoint), multiplied by
    act'(y) * gain, written as a split image only (its consumers are the data- and weight-gradient kernels), with the bias
    gradient's per-block channel sums reduced inside the launch and column-summed afterwards (two deterministic stages).
    Replaces upfirdn2d -> rick_bias_act_bwd: 12 instead of 20 bytes per element.  |result| <= gain * max(1, slope) * max|g|
    (taps >= 0, sum 1).  -> (image, gb or None)"""
    n, c, h, w = g.shape
    kh, kw = taps.shape
    oh, ow = h + pad4[2] + pad4[3] - kh + 1, w + pad4[0] + pad4[1] - kw + 1
    assert (n, c, oh, ow) == tuple(y.shape)
    coef = abs(gain) * max(1.0, abs(slope))
    img = sp.SplitImage(torch.empty_like(y), sp.new_words(4, g.device), (amax_g, None, coef))
    ex = SplitOut()
    ex.split_out, ex.split_hdr, ex.bound0, ex.bound1, ex.bound_coef = ptr(img.data), ptr(img.hdr), ptr(amax_g), None, coef
    ex.no_f32 = 1
    ex.adj_ref, ex.adj_slope, ex.adj_gain = ptr(y), float(slope), float(gain)
    part = None
    if want_b:
        rows = lib.rick_upfirdn2d_adjoint_rows(n, oh, ow)
        part = torch.empty(rows * c, device=g.device, dtype=torch.float32)
        ex.adj_partials = ptr(part)
    from .conv import hbm_launch
    check(hbm_launch('upfirdn2d', 4 * (g.numel() + 2 * y.numel()), lib.rick_upfirdn2d_ex_f32, ptr(g), ptr(taps), None, n, h, w, c, kh, kw,
                     1, 1, 1, 1, pad4[0], pad4[1], pad4[2], pad4[3], None, ctypes.byref(ex), stream_ptr()), 'rick_upfirdn2d_ex_f32')
    gb = None
    if want_b:
        sunk = sink_b is not None
        gb = sink_b if sunk else torch.empty(c, device=g.device, dtype=torch.float32)
        if rows > 512 and rows % 64 == 0:      # two stages: [rows / 64][64 * c] -> [64][c] -> [c] (a single stage would walk 8 K rows per thread)
            mid = torch.empty(64 * c, device=g.device, dtype=torch.float32)
            check(lib.rick_colsum_f32(ptr(part), ptr(mid), rows // 64, 64 * c, 64 * c, 0, stream_ptr()), 'rick_colsum_f32')
            from .fused_act import defer_colsum
            if not (sunk and defer_colsum(mid, gb, 64, c, c)):
                check(lib.rick_colsum_f32(ptr(mid), ptr(gb), 64, c, c, int(sunk), stream_ptr()), 'rick_colsum_f32')
        else:
            from .fused_act import defer_colsum
            if not (sunk and defer_colsum(part, gb, rows, c, c)):
                check(lib.rick_colsum_f32(ptr(part), ptr(gb), rows, c, c, int(sunk), stream_ptr()), 'rick_colsum_f32')
        if sunk:
            gb = None
    return img, gb


def _amax_epilogue(word, bias=None, slope=0.2, gain=1.0, act=False):
    e = _epilogue(bias, None, None, slope, gain)
    if not act:
        e.act = 0
    e.amax = ptr(word)
    return e


_supported = {}


def block_supported(x, w1, w2, ws):
    """Every MFMA launch of the block (forward, data and weight gradients) has a split-image form for these shapes."""
    N, C, H, W = x.shape
    O = w2.shape[0]
    key = (N, C, H, W, O)
    ok = _supported.get(key)
    if ok is None:
        ok = (C % 128 == 0 and O % 128 == 0 and H % 2 == 0 and W % 2 == 0 and w1.shape == (C, C, 3, 3) and w2.shape == (O, C, 3, 3)
              and ws.shape == (O, C, 1, 1) and N <= 65535)
        if ok:
            t3 = [(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
            t3s = [(ky, kx, ky * 3 + kx) for ky in range(3) for kx in range(3)]
            t3T = [(1 - ky, 1 - kx, ky * 3 + kx) for ky in range(3) for kx in range(3)]
            H2, W2 = H // 2, W // 2
            gs = [(_geom(N, H, W, C, H, W, C, H, W, 1, 1, 0, 0, t3, 9), 'iw'),                       # conv1 + its wgrad
                  (_geom(N, H, W, C, H, W, C, H, W, 1, 1, 0, 0, t3T, 9), 'i'),                       # conv1 dgrad
                  (_geom(N, H + 1, W + 1, C, H2, W2, O, H2, W2, 2, 1, 0, 0, t3s, 9), 'iw'),         # conv2 + its wgrad
                  (_geom(N, H2, W2, C, H2, W2, O, H2, W2, 1, 1, 0, 0, [(0, 0, 0)], 1), 'iw'),       # skip conv + its wgrad
                  (_geom(N, H2, W2, O, H2, W2, C, H2, W2, 1, 1, 0, 0, [(0, 0, 0)], 1), 'i')]        # skip dgrad
            for g, kinds in gs:
                if 'i' in kinds and not lib.rick_conv_igemm_split_supported(ctypes.byref(g)):
                    ok = False
                if 'w' in kinds and not lib.rick_conv_wgrad_split_supported(ctypes.byref(g)):
                    ok = False
            if lib.rick_convt2_workspace_bytes(N, H2, W2, O, C, H + 1, W + 1) < 0:
                ok = False
        _supported[key] = ok
    return ok


class _DResBlock(Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, ws, taps, cfg):
        sc1, sc2, scs, slope, gain, pad2, pads, k1, k2, ks, compose = cfg
        N, C, H, W = x.shape
        O = w2.shape[0]
        xpk = sp.taken(x, '_rick_split')
        xc = x.contiguous(memory_format=torch.channels_last)
        if xpk is None:
            xpk = sp.split_pack(xc)                     # block input from a producer that is not fused (the first block)
        dev = x.device
        A1, A2, A3 = sp.new_amax(dev), sp.new_amax(dev), sp.new_amax(dev)
        b1c, b2c = b1.contiguous(), b2.contiguous()
        t1 = _conv_launch(None, _pack(w1, sc1, (k1, 'w/conv')), C, 3, 3, 1, 1, epi=_amax_epilogue(A1, b1c, slope, gain, True), x_split=xpk)
        # blur(t1): only conv2 and its weight gradient read it -> split image only; |blur| <= max |t1| (taps >= 0, sum 1)
        _, b1pk = _fir_ex(t1, taps, 1, 1, (pad2[0], pad2[1], pad2[0], pad2[1]), split_bound=A1, no_f32=True)
        t2 = _conv_launch(None, _pack(w2, sc2, (k2, 'w/conv')), O, 3, 3, 2, 0, epi=_amax_epilogue(A2, b2c, slope, gain, True), x_split=b1pk)
        # skip path: the FIR evaluated at the even positions only (models.ResBlock._skip), bound = the input image's own
        xb = xpk.bound
        _, xspk = _fir_ex(xc, taps, 1, 2, (pads[0], pads[1], pads[0], pads[1]), split_bound=xb[0], bound1=xb[1], coef=xb[2], no_f32=True)
        sk = _conv_launch(None, _pack(ws, scs, (ks, 'w/conv')), O, 1, 1, 1, 0, epi=_amax_epilogue(A3), x_split=xspk)
        out = torch.empty_like(t2)
        opk = sp.SplitImage(torch.empty_like(t2), sp.new_words(4, dev), (A2, A3, _SQ))
        check(lib.rick_add_scale_split_f32(ptr(t2), ptr(sk), ptr(out), ptr(opk.data), ptr(opk.hdr), ptr(A2), ptr(A3),
                                           t2.numel() // O, O, _SQ, stream_ptr()), 'rick_add_scale_split_f32')
        sp.hand(out, '_rick_split', opk)
        ctx.save_for_backward(x, w1, b1, w2, b2, ws, taps, t1, t2, xpk.data, xpk.hdr, b1pk.data, b1pk.hdr, xspk.data, xspk.hdr)
        ctx.cfg = cfg
        ctx.sink = grad_sink_enabled()
        ctx.plike = (False, param_like(w1), param_like(b1), param_like(w2), param_like(b2), param_like(ws))
        ctx.set_materialize_grads(False)
        return out, t1, t2

    @staticmethod
    def backward(ctx, g_out, g_t1, g_t2):
        x, w1, b1, w2, b2, ws, taps, t1, t2, xd, xh, bd, bh, sd, sh = ctx.saved_tensors
        sc1, sc2, scs, slope, gain, pad2, pads, k1, k2, ks, compose = ctx.cfg
        if torch.is_grad_enabled():             # create_graph=True: the per-layer, twice-differentiable composition
            from ._twice import second_order_backward
            res = second_order_backward(compose, (x, w1, b1, w2, b2, ws), ctx.needs_input_grad[:6], (g_out, g_t1, g_t2), ctx.plike)
            return (*res, None, None)
        N, C, H, W = x.shape
        O = w2.shape[0]
        dev = x.device
        need = [n and not skip_param_grad(p) for n, p in zip(ctx.needs_input_grad[:6], ctx.plike)]
        need_x = ctx.needs_input_grad[0]
        xpk, b1pk, xspk = sp.SplitImage(xd, xh), sp.SplitImage(bd, bh), sp.SplitImage(sd, sh)
        gw1 = gb1 = gw2 = gb2 = gws = gx = None
        flip = _flipped(taps)
        if g_out is None and g_t2 is None and g_t1 is None:
            return (None,) * 8
        if g_out is None:
            g_out = torch.zeros_like(t2)
        g_out = g_out.contiguous(memory_format=torch.channels_last)
        # ---- conv2 branch + the skip branch's incoming gradient, from ONE read of g_out
        mul2 = _SQ
        if g_t2 is not None:                    # a loss on the returned feature map: fold it in (rare; plain tensor ops)
            g2in = torch.add(g_t2, g_out, alpha=_SQ).contiguous(memory_format=torch.channels_last)
            A = sp.amax(g2in)
            gz2, _, gb2 = _act_adjoint_split(g2in, t2, slope, gain, A, None, need[4], param_sink(b2, O, ctx.sink and need[4]))
            Ag = sp.taken(g_out, '_rick_amax')
            if Ag is None:
                Ag = sp.amax(g_out)
            gsk = sp.split_pack(g_out * _SQ, Ag, None, _SQ)
        else:
            Ag = sp.taken(g_out, '_rick_amax')
            if Ag is None:
                Ag = sp.amax(g_out)
            gz2, gsk, gb2 = _act_adjoint_split(g_out, t2, slope, gain * _SQ, Ag, mul2, need[4], param_sink(b2, O, ctx.sink and need[4]))
        if need[3]:
            gw2 = _wgrad_launch(None, None, 3, 3, 2, 0, sc2, out=_sink_target((k2, 'w'), w2.shape, ctx.sink), a_split=gz2, b_split=b1pk)
        upstream = need_x or need[1] or need[2]
        if upstream:
            A1g = sp.new_amax(dev)
            g_b1 = _convT_launch(None, _pack(w2.transpose(0, 1), sc2, (k2, 'w/T/convT')), C, 3, 3, 2, 0, (H + 1, W + 1), x_split=gz2,
                                 amax=A1g)
            kh = taps.shape[0]
            adj2 = (kh - pad2[0] - 1, W - (W + 1) + pad2[0], kh - pad2[0] - 1, H - (H + 1) + pad2[0])   # op/upfirdn2d.py:111-114
            if g_t1 is None and C % 64 == 0:
                # blur adjoint and activation adjoint in one launch: the blurred gradient never exists as a tensor
                gz1, gb1 = _fir_adjoint_split(g_b1, flip, adj2, t1, slope, gain, A1g, need[2], param_sink(b1, C, ctx.sink and need[2]))
            else:
                A1f = sp.new_amax(dev)
                g_t1b, _ = _fir_ex(g_b1, flip, 1, 1, adj2, amax=A1f)
                if g_t1 is not None:                # a loss on the returned feature map (rare): plain tensor ops
                    g_t1b = g_t1b + g_t1
                    A1f = sp.amax(g_t1b)
                gz1, _, gb1 = _act_adjoint_split(g_t1b, t1, slope, gain, A1f, None, need[2], param_sink(b1, C, ctx.sink and need[2]))
            if need[1]:
                gw1 = _wgrad_launch(None, None, 3, 3, 1, 1, sc1, out=_sink_target((k1, 'w'), w1.shape, ctx.sink), a_split=gz1, b_split=xpk)
            if need_x:
                gx = _convT_launch(None, _pack(w1.transpose(0, 1), sc1, (k1, 'w/T/convT')), C, 3, 3, 1, 1, (H, W), x_split=gz1)
        if need[5]:
            gws = _wgrad_launch(None, None, 1, 1, 1, 0, scs, out=_sink_target((ks, 'w'), ws.shape, ctx.sink), a_split=gsk, b_split=xspk)
        if need_x:
            g_xs = _convT_launch(None, _pack(ws.transpose(0, 1), scs, (ks, 'w/T/convT')), C, 1, 1, 1, 0, (H // 2, W // 2), x_split=gsk)
            # adjoint of the decimating FIR = zero-insertion upsampling with the flipped taps, ADDED into the main path's gradient
            kh = taps.shape[0]
            adjs = (kh - pads[0] - 1, W - (W // 2) * 2 + pads[0], kh - pads[0] - 1, H - (H // 2) * 2 + pads[0])
            Agx = sp.new_amax(dev)
            gx, _ = _fir_ex(g_xs, flip, 2, 1, adjs, out=gx, amax=Agx, accumulate=True)
            sp.hand(gx, '_rick_amax', Agx)
        return gx, gw1, gb1, gw2, gb2, gws, None, None


def d_resblock(x, w1, b1, w2, b2, ws, taps, cfg):
    return _DResBlock.apply(x, w1, b1, w2, b2, ws, taps, cfg)


class _DInput(Function):
    """The discriminator's input layer (model_probe_tune.py:679-680: 1x1 EqualConv2d 3 -> C without bias, FusedLeakyReLU) as one
    launch that writes the activation as fp32 and as the split image the first ResBlock's convolutions read — bit-identical to
    thin_bwdx -> fused_leaky_relu, without the 3 extra passes over the 128-channel map (and the stand-alone maximum + pack passes
    the first block would otherwise need).  Bound of the image: gain * (J * max |W| * max |img| + max |b|)."""

    @staticmethod
    def forward(ctx, img, weight, bias, wscale, slope, gain, compose):
        O, J = weight.shape[0], weight.shape[1]
        n, _, h, w = img.shape
        t = img.contiguous()
        W = (weight.view(O, J) * wscale).t().contiguous()           # [J, C], the per-layer path's own expression
        bc = bias.contiguous()
        x = torch.empty((n, O, h, w), device=img.device, dtype=torch.float32, memory_format=torch.channels_last)
        pk = sp.SplitImage(torch.empty_like(x), sp.new_words(4, img.device))
        a_img = sp.amax(t)
        # (a tensor of its own: an in-place write into an arena slice would bump the version counter every saved header shares)
        bound = torch.nn.functional.pad((gain * (J * W.abs().max() * sp.amax_value(a_img) + bc.abs().max())).reshape(1),
                                        (0, sp.AMAX_FLOATS - 1))
        pk.bound = (bound, None, 1.0)
        ex = SplitOut()
        ex.split_out, ex.split_hdr, ex.bound0, ex.bound1, ex.bound_coef = ptr(pk.data), ptr(pk.hdr), ptr(bound), None, 1.0
        check(lib.rick_d_input_f32(ptr(t), ptr(W), ptr(bc), ptr(x), n, h * w, O, J, float(slope), float(gain), ctypes.byref(ex),
                                   stream_ptr()), 'rick_d_input_f32')
        sp.hand(x, '_rick_split', pk)
        ctx.save_for_backward(img, weight, bias, x, W)
        ctx.cfg = (wscale, slope, gain, compose)
        ctx.sink = grad_sink_enabled()
        ctx.plike = (False, param_like(weight), param_like(bias))
        return x

    @staticmethod
    def backward(ctx, g):
        from .fused_act import _ActAdjoint
        from .misc import _ThinFwd, _ThinWgrad
        img, weight, bias, x, W = ctx.saved_tensors
        wscale, slope, gain, compose = ctx.cfg
        if torch.is_grad_enabled():
            from ._twice import second_order_backward
            res = second_order_backward(compose, (img, weight, bias), ctx.needs_input_grad[:3], g, ctx.plike)
            return (*res, None, None, None, None)
        O, J = weight.shape[0], weight.shape[1]
        need_w = ctx.needs_input_grad[1] and not skip_param_grad(ctx.plike[1])
        need_b = ctx.needs_input_grad[2] and not skip_param_grad(ctx.plike[2])
        gz, gb, _ = _ActAdjoint.apply(g, x, None, slope, gain, need_b, False, param_sink(bias, O, ctx.sink and need_b))
        gimg = gw = None
        if ctx.needs_input_grad[0]:
            gimg = _ThinFwd.apply(gz, W.unsqueeze(0))
        if need_w:
            gw = (_ThinWgrad.apply(img, gz).sum(0).t() * wscale).reshape(weight.shape)
        return gimg, gw, gb, None, None, None, None


def d_input(img, weight, bias, wscale, slope, gain, compose):
    return _DInput.apply(img, weight, bias, float(wscale), float(slope), float(gain), compose)
