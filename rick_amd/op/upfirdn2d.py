"""upfirdn2d on MI355X — drop-in for the reference's ``op.upfirdn2d``
(op/upfirdn2d.py:145-156): ``upfirdn2d(input[N,C,H,W], kernel[kh,kw], up=1, down=1, pad=(p0,p1))``.

The operator is linear in ``input`` and its adjoint is again an upfirdn2d (flipped FIR, up and
down exchanged, the gradient pads of op/upfirdn2d.py:111-114), so ONE autograd node type,
closed under differentiation, gives gradients of every order; the reference reaches the same
result with a Function / Backward-Function pair (op/upfirdn2d.py:19-142).  Every launch is
``rick_upfirdn2d_f32`` (rick_amd/csrc/upfirdn2d.hip).

Layout: inputs with C % 64 == 0 (or C < 64 and C % 4 == 0) run channels-last through the NHWC
kernel (major = N, minor = C); anything else (e.g. the 3-channel RGB skip) runs planar
(major = N*C, minor = 1).  The result is returned in the layout that ran.
"""
import torch
import ctypes

from torch.autograd import Function

from .._lib import check, lib, ptr, require_cuda_f32, stream_ptr


def _fir(x, taps, up_xy, down_xy, pad4):
    """One kernel launch.  pad4 = (x0, x1, y0, y1).  Output extent per axis:
    (in*up + pad0 + pad1 - k)//down + 1 (op/upfirdn2d.py:103-104)."""
    n, c, h, w = x.shape
    kh, kw = taps.shape
    oh = (h * up_xy[1] + pad4[2] + pad4[3] - kh) // down_xy[1] + 1
    ow = (w * up_xy[0] + pad4[0] + pad4[1] - kw) // down_xy[0] + 1
    if oh <= 0 or ow <= 0:
        raise RuntimeError(f'upfirdn2d: empty output ({oh}x{ow})')
    if x.dtype != torch.float32:
        # float64 / float16 (the reference extension's other dtypes): the generic planar kernel
        from .._lib import DTYPE_CODE
        x = x.contiguous()
        y = torch.empty((n, c, oh, ow), device=x.device, dtype=x.dtype)
        check(lib.rick_upfirdn2d_any(ptr(x), ptr(taps), ptr(y), DTYPE_CODE[x.dtype], n * c, h, w, kh, kw, up_xy[0], up_xy[1],
                                     down_xy[0], down_xy[1], pad4[0], pad4[1], pad4[2], pad4[3], stream_ptr()), 'rick_upfirdn2d_any')
        return y
    if c % 64 == 0 or (c < 64 and c % 4 == 0):
        x = x.contiguous(memory_format=torch.channels_last)
        y = torch.empty((n, c, oh, ow), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
        major, minor = n, c
    else:
        x = x.contiguous()
        y = torch.empty((n, c, oh, ow), device=x.device, dtype=x.dtype)
        major, minor = n * c, 1
    from .conv import hbm_launch
    check(hbm_launch('upfirdn2d', 4 * (x.numel() + y.numel()), lib.rick_upfirdn2d_f32, ptr(x), ptr(taps), ptr(y), major, h, w, minor,
                     kh, kw, up_xy[0], up_xy[1], down_xy[0], down_xy[1], pad4[0], pad4[1], pad4[2], pad4[3], stream_ptr()),
          'rick_upfirdn2d_f32')
    return y


def _abs_sum(taps):
    """sum |taps| as a host float, cached on the tensor (one sync at the first, eager, use: never under graph capture)."""
    ent = getattr(taps, '_rick_abs_sum', None)
    if ent is None or ent[0] != taps._version:
        ent = (taps._version, float(taps.abs().sum()))
        taps._rick_abs_sum = ent
    return ent[1]


def _flipped(taps):
    """flip(taps) for the adjoint, cached ON the FIR tensor object (the networks' filters are constant buffers:
    without the cache every blur backward launches a flip and a copy of a 16-element tensor).  The flipped tensor
    points back at its source, so double backward reuses the pair."""
    ent = getattr(taps, '_rick_flipped', None)
    if ent is None or ent[0] != taps._version:
        f = torch.flip(taps, [0, 1]).contiguous()
        f._rick_flipped = (f._version, taps)
        ent = (taps._version, f)
        taps._rick_flipped = ent
    return ent[1]


class _UpFirDn(Function):
    """y = upfirdn2d(x; taps, up, down, pad).  backward(g) = _UpFirDn(g; flip(taps), down, up, adj_pad)."""

    @staticmethod
    def forward(ctx, x, taps, up_xy, down_xy, pad4):
        y = _fir(x, taps, up_xy, down_xy, pad4)
        kh, kw = taps.shape
        h, w = x.shape[2], x.shape[3]
        oh, ow = y.shape[2], y.shape[3]
        # pads of the adjoint operator (same expressions as op/upfirdn2d.py:111-114)
        adj = (kw - pad4[0] - 1, w * up_xy[0] - ow * down_xy[0] + pad4[0] - up_xy[0] + 1,
               kh - pad4[2] - 1, h * up_xy[1] - oh * down_xy[1] + pad4[2] - up_xy[1] + 1)
        ctx.flipped = _flipped(taps)         # a constant of the op (no gradient flows to the FIR taps)
        ctx.adjoint = (down_xy, up_xy, adj)
        return y

    @staticmethod
    def backward(ctx, g):
        a_up, a_down, a_pad = ctx.adjoint
        gx = _UpFirDn.apply(g, ctx.flipped, a_up, a_down, a_pad)
        return gx, None, None, None, None


def _host_route(x, taps, up, down, pad):
    """CPU tensors (op/upfirdn2d.py:145-149 sends them to its torch-native form: ADA in a data worker, model debugging on
    the host).  Plain differentiable torch, any float dtype: the zero-stuffed, padded (negative pad: cropped) signal is
    written into one zero canvas by a strided slice assignment, correlated with the flipped taps as a 1-channel
    convolution over [N*C, 1, H', W'] — the arithmetic the reference's CPU path performs, so fp32 results are equal
    bit for bit — and decimated by a strided slice.  This is host-side product code, not the test oracle."""
    import torch.nn.functional as F
    n, c, h, w = x.shape
    kh, kw = taps.shape
    ch, cw = h * up + pad[0] + pad[1], w * up + pad[0] + pad[1]
    oh, ow = (ch - kh) // down + 1, (cw - kw) // down + 1
    if ch < kh or cw < kw or oh <= 0 or ow <= 0:
        raise RuntimeError(f'upfirdn2d: empty output ({oh}x{ow})')

    def span(extent, canvas):
        # input sample i sits at canvas position pad0 + i * up: the samples whose position is inside [0, canvas)
        lo = max(0, -(pad[0] // up))                    # smallest i with pad0 + i * up >= 0
        hi = min(extent, (canvas - 1 - pad[0]) // up + 1)
        return lo, hi, pad[0] + lo * up
    y0, y1, py = span(h, ch)
    x0, x1, px = span(w, cw)
    canvas = x.new_zeros((n * c, 1, ch, cw))
    if y1 > y0 and x1 > x0:
        canvas[:, 0, py:py + (y1 - y0 - 1) * up + 1:up, px:px + (x1 - x0 - 1) * up + 1:up] = \
            x.reshape(n * c, h, w)[:, y0:y1, x0:x1]
    full = F.conv2d(canvas, torch.flip(taps, [0, 1]).to(x.dtype).view(1, 1, kh, kw))
    return full[:, 0, ::down, ::down].reshape(n, c, oh, ow)


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """Same signature and semantics as the reference op (op/upfirdn2d.py:145-156): one up / down / pad pair for both
    axes; device tensors run the HIP kernels, CPU tensors the torch route above (the reference's own device dispatch)."""
    if input.ndim != 4 or kernel.ndim != 2:
        raise RuntimeError('upfirdn2d expects input [N,C,H,W] and kernel [kh,kw]')
    if input.device.type == 'cpu':
        return _host_route(input, kernel, int(up), int(down), (int(pad[0]), int(pad[1])))
    from .._lib import require_cuda_float
    require_cuda_float(input, kernel)      # float16 / float32 / float64 like the reference extension (one dtype per call)
    return _UpFirDn.apply(input, kernel.contiguous(), (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))


class _FirAct(Function):
    """upfirdn2d (4x4 taps, up = down = 1, channels-last) followed by NoiseInjection + bias + LeakyReLU in the SAME
    launch (rick_upfirdn2d_act_f32): the blur after an upsampling StyledConv and its activation tail
    (model_probe_tune.py:263-268, 343-346) without the intermediate feature map round trip.  Bit-identical to
    upfirdn2d -> fused_noise_bias_act.  Hand-written first-order backward; under create_graph=True the backward
    differentiates that two-op form (op/_twice.py)."""

    @staticmethod
    def forward(ctx, x, taps, pad4, bias, noise, nw, slope, gain, extra=None):
        """extra = (next_s, bwd_scale) or None.  next_s [N, C]: also write the output as the split-image operand of the next
        modulated convolution (its style folded in; needs the bound `x._rick_bound` the transposed convolution attached).
        bwd_scale [N, C]: the demodulation scale of the convolution BEFORE this blur — the backward writes the blur's adjoint
        as the image of bwd_scale * gradient for that convolution's data / weight gradient kernels."""
        from .conv import _epilogue, grad_sink_enabled
        from .._lib import SplitOut
        ctx.params = (bias, nw, grad_sink_enabled())      # op.grad_sink(): bias / noise-strength gradients go straight to .grad
        n, c, h, w = x.shape
        kh, kw = taps.shape
        oh, ow = h + pad4[2] + pad4[3] - kh + 1, w + pad4[0] + pad4[1] - kw + 1
        x_in = x
        x = x.contiguous(memory_format=torch.channels_last)
        bias = bias.contiguous()
        noise = noise.contiguous()
        nw = nw.contiguous()
        y = torch.empty((n, c, oh, ow), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
        tail = _epilogue(bias, noise, nw, slope, gain)
        next_s, ctx.bwd_scale = extra if extra is not None else (None, None)
        from . import split as sp
        xb = sp.taken(x_in, '_rick_bound')
        if xb is not None and c % 4 == 0:
            from . import split as sp
            from .modconv import _bound_tail, _tensor_amax
            ex = SplitOut()
            A = sp.new_amax(x.device)
            ex.amax = ptr(A)
            img = None
            if next_s is not None:
                ns = next_s.contiguous()
                # |y * next_s| <= max|next_s| * gain * (sum|taps| * bound(x) + |nw| * max|noise| + max|bias|)
                bound = _bound_tail(xb[0], xb[1] * _abs_sum(taps), gain, nw, _tensor_amax(noise), bias, ns)
                img = sp.SplitImage(torch.empty_like(y), sp.new_words(4, x.device), (bound, None, 1.0))
                img.scale_of = next_s
                ex.split_out, ex.split_hdr, ex.bound0, ex.bound1, ex.bound_coef, ex.chan_scale = (
                    ptr(img.data), ptr(img.hdr), ptr(bound), None, 1.0, ptr(ns))
            check(lib.rick_upfirdn2d_ex_f32(ptr(x), ptr(taps), ptr(y), n, h, w, c, kh, kw, 1, 1, 1, 1, pad4[0], pad4[1], pad4[2],
                                            pad4[3], ctypes.byref(tail), ctypes.byref(ex), stream_ptr()), 'rick_upfirdn2d_ex_f32')
            sp.hand(y, '_rick_amax', A)
            if img is not None:
                sp.hand(y, '_rick_split', img)
        else:
            from .conv import hbm_launch
            check(hbm_launch('upfirdn2d', 4 * (x.numel() + y.numel()), lib.rick_upfirdn2d_act_f32, ptr(x), ptr(taps), ptr(y), n, h, w,
                             c, kh, kw, 1, 1, 1, 1, pad4[0], pad4[1], pad4[2], pad4[3], ctypes.byref(tail), stream_ptr()),
                  'rick_upfirdn2d_act_f32')
        ctx.save_for_backward(y, noise, x_in, taps)       # (x only for the create_graph route; no copy: it is the op's input)
        ctx.pad4 = pad4
        ctx.flipped = _flipped(taps)
        ctx.cfg = (slope, gain, (kw - pad4[0] - 1, w - ow + pad4[0], kh - pad4[2] - 1, h - oh + pad4[2]))
        return y


    @staticmethod
    def backward(ctx, g):
        from .fused_act import _ActAdjoint, fused_noise_bias_act, param_sink
        y, noise, x_in, taps = ctx.saved_tensors
        slope, gain, adj = ctx.cfg
        bias, nw, sink = ctx.params
        if torch.is_grad_enabled():     # create_graph=True: blur -> activation from the twice-differentiable ops
            from ._twice import second_order_backward
            pad4 = ctx.pad4
            gx, gb, gw = second_order_backward(
                lambda x_, b_, nw_: fused_noise_bias_act(_UpFirDn.apply(x_, taps, (1, 1), (1, 1), pad4), b_, noise, nw_, slope, gain),
                (x_in, bias, nw), [ctx.needs_input_grad[i] for i in (0, 3, 5)], g)
            return (gx, None, None, gb, None, gw, None, None, None)
        want_b, want_w = ctx.needs_input_grad[3], ctx.needs_input_grad[5]
        gz, gb, gw = _ActAdjoint.apply(g, y, noise, slope, gain, want_b, want_w,
                                       param_sink(bias, y.shape[1], sink and want_b), param_sink(nw, 1, sink and want_w))
        gx = None
        if ctx.needs_input_grad[0]:
            bs = ctx.bwd_scale
            if bs is not None and y.shape[1] % 4 == 0:
                # the blur's adjoint as fp32 and as the image of bwd_scale * gradient (the transposed convolution's backward
                # operand): |.| <= sum|taps| * gain * max(1, slope) * max|g| * max|bwd_scale|
                from . import split as sp
                from .dblock import _fir_ex
                from .modconv import _bound_tail, _tensor_amax
                bsc = bs.contiguous()
                bound = _bound_tail(_tensor_amax(g), _abs_sum(ctx.flipped) * abs(gain) * max(1.0, abs(slope)), 1.0, mul=bsc)
                gx, img = _fir_ex(gz, ctx.flipped, 1, 1, adj, split_bound=bound, chan_scale=bsc)
                img.scale_of = bs
                sp.hand(gx, '_rick_split', img)
            else:
                gx = _fir(gz, ctx.flipped, (1, 1), (1, 1), adj)
        return (gx, None, None, gb, None, gw, None, None, None)


def upfirdn2d_noise_bias_act(input, kernel, pad, bias, noise, noise_weight, negative_slope=0.2, gain=2 ** 0.5, next_s=None,
                             bwd_scale=None):
    """gain * lrelu(upfirdn2d(input, kernel, pad=pad) + bias + noise_weight * noise) — one launch when the fused
    path applies (4x4 taps, C % 64 == 0), the two separate ops otherwise.  next_s / bwd_scale: split-image hand-over to the
    neighbouring modulated convolutions (see _FirAct.forward)."""
    require_cuda_f32(input, kernel, bias, noise, noise_weight)
    if kernel.shape == (4, 4) and input.shape[1] % 64 == 0 and input.shape[0] <= 65535:
        extra = (next_s, bwd_scale) if (next_s is not None or bwd_scale is not None) else None
        return _FirAct.apply(input, kernel.contiguous(), (pad[0], pad[1], pad[0], pad[1]), bias, noise, noise_weight,
                             float(negative_slope), float(gain), extra)
    from .fused_act import fused_noise_bias_act
    return fused_noise_bias_act(upfirdn2d(input, kernel, pad=pad), bias, noise, noise_weight, negative_slope, gain)
