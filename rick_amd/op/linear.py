"""EqualLinear with gradients on a short batch — ``F.linear(input, weight * scale, bias * lr_mul)`` of the reference
(model_probe_tune.py:157-168) for the discriminator's final layers (:699-702), which sit in every D pass of the loop.

Three products closed under differentiation (rick_amd/csrc/linear.hip), each ONE pass over the [O, K] matrix with a fixed
summation order:

    P1(x, W, b) = a * x W^T + m * b     bwd(g): (P2(g, W), P3(g, x), m * sum_b g)
    P2(g, W)    = a * g W               bwd(h): (P1(h, W),  P3(g, h))
    P3(g, x)    = a * g^T x             bwd(H): (P1(x, H),  P2(g, H))

so R1's double backward stays on the same three kernels.  Replaces torch.addmm (rocBLAS / hipBLASLt: several launches per
layer and, for M = batch, K = 8192, solutions the library is free to split over K)."""
import torch
from torch.autograd import Function

from .._lib import check, lib, ptr, require_cuda_f32, stream_ptr

MAX_BATCH = 16


def supported(x, weight):
    return (x.ndim == 2 and x.is_cuda and x.dtype == torch.float32 and 1 <= x.shape[0] <= MAX_BATCH and x.shape[1] % 4 == 0
            and weight.ndim == 2 and weight.shape[1] == x.shape[1])


def _ws(n, like):
    return torch.empty(n, device=like.device, dtype=torch.float32) if n > 0 else None


def _p1(x, W, bias, alpha, bias_mul):
    x, W = x.contiguous(), W.contiguous()
    B, K = x.shape
    O = W.shape[0]
    out = torch.empty((B, O), device=x.device, dtype=torch.float32)
    ws = _ws(lib.rick_linear_fwd_workspace_floats(B, K, O), x)
    b = bias.contiguous() if bias is not None else None
    check(lib.rick_linear_fwd_f32(ptr(x), ptr(W), ptr(b), ptr(out), B, K, O, alpha, bias_mul, ptr(ws), stream_ptr()),
          'rick_linear_fwd_f32')
    return out


def _p2(g, W, alpha):
    g, W = g.contiguous(), W.contiguous()
    B, O = g.shape
    K = W.shape[1]
    out = torch.empty((B, K), device=g.device, dtype=torch.float32)
    ws = _ws(lib.rick_linear_dgrad_workspace_floats(B, K, O), g)
    check(lib.rick_linear_dgrad_f32(ptr(g), ptr(W), ptr(out), B, K, O, alpha, ptr(ws), stream_ptr()), 'rick_linear_dgrad_f32')
    return out


def _p3(g, x, alpha, bias_mul=1.0, want_w=True, want_b=False, sink_w=None, sink_b=None):
    """-> (gw or None, gb or None); with sinks the sums are ADDED into them and nothing is returned for that operand."""
    g, x = g.contiguous(), x.contiguous()
    B, O = g.shape
    K = x.shape[1]
    if (want_w and sink_w is None) or (want_b and sink_b is None):
        sink_w = sink_b = None                     # one accumulate switch serves both
    sunk = (want_w and sink_w is not None) or (want_b and sink_b is not None)
    gw = (sink_w if sunk else torch.empty((O, K), device=g.device, dtype=torch.float32)) if want_w else None
    gb = (sink_b if sunk else torch.empty(O, device=g.device, dtype=torch.float32)) if want_b else None
    if gw is not None or gb is not None:
        check(lib.rick_linear_wgrad_f32(ptr(g), ptr(x), ptr(gw), ptr(gb), B, K, O, alpha, bias_mul, int(sunk), stream_ptr()),
              'rick_linear_wgrad_f32')
    return (None, None) if sunk else (gw, gb)


class _P1(Function):
    @staticmethod
    def forward(ctx, x, W, bias, alpha, bias_mul):
        from .conv import grad_sink_enabled, param_like
        ctx.save_for_backward(x, W)
        ctx.cfg = (alpha, bias_mul)
        ctx.params = (W, bias, grad_sink_enabled())       # the switch as the forward's thread sees it (op/conv.py)
        ctx.plike = (param_like(W), param_like(bias))
        return _p1(x, W, bias, alpha, bias_mul)

    @staticmethod
    def backward(ctx, g):
        from .conv import skip_param_grad
        from .fused_act import param_sink
        x, W = ctx.saved_tensors
        alpha, bias_mul = ctx.cfg
        Wp, bp, sink = ctx.params
        want_w = ctx.needs_input_grad[1] and not skip_param_grad(ctx.plike[0])
        want_b = bp is not None and ctx.needs_input_grad[2] and not skip_param_grad(ctx.plike[1])
        gx = gw = gb = None
        if torch.is_grad_enabled():                       # create_graph=True: stay inside the family
            if ctx.needs_input_grad[0]:
                gx = _P2.apply(g, W, alpha)
            if want_w:
                gw = _P3.apply(g, x, alpha)
            if want_b:
                gb = g.sum(0) * bias_mul
            return gx, gw, gb, None, None
        if ctx.needs_input_grad[0]:
            gx = _p2(g, W, alpha)
        if want_w or want_b:
            sink = sink and not torch.is_grad_enabled()
            gw, gb = _p3(g, x, alpha, bias_mul, want_w, want_b,
                         param_sink(Wp, W.numel(), sink and want_w), param_sink(bp, W.shape[0], sink and want_b))
            if gw is not None and gw.shape != Wp.shape:
                gw = gw.view(Wp.shape)
        return gx, gw, gb, None, None


class _P2(Function):
    @staticmethod
    def forward(ctx, g, W, alpha):
        from .conv import param_like
        ctx.save_for_backward(g, W)
        ctx.alpha, ctx.plike = alpha, param_like(W)
        return _p2(g, W, alpha)

    @staticmethod
    def backward(ctx, h):
        from .conv import skip_param_grad
        g, W = ctx.saved_tensors
        gg = gw = None
        if ctx.needs_input_grad[0]:
            gg = _P1.apply(h, W, None, ctx.alpha, 0.0)
        if ctx.needs_input_grad[1] and not skip_param_grad(ctx.plike):
            gw = _P3.apply(g, h, ctx.alpha)
        return gg, gw, None


class _P3(Function):
    @staticmethod
    def forward(ctx, g, x, alpha):
        ctx.save_for_backward(g, x)
        ctx.alpha = alpha
        return _p3(g, x, alpha)[0]

    @staticmethod
    def backward(ctx, H):
        g, x = ctx.saved_tensors
        gg = gx = None
        if ctx.needs_input_grad[0]:
            gg = _P1.apply(x, H, None, ctx.alpha, 0.0)
        if ctx.needs_input_grad[1]:
            gx = _P2.apply(g, H, ctx.alpha)
        return gg, gx, None


def linear(x, weight, bias=None, alpha=1.0, bias_mul=1.0):
    """alpha * x @ weight^T + bias_mul * bias for x [B <= 16, K % 4 == 0] (any order of autograd)."""
    require_cuda_f32(x, weight, bias)
    if not supported(x, weight):
        raise RuntimeError(f'op.linear: needs x [B <= {MAX_BATCH}, K % 4 == 0] and weight [O, K]; got {tuple(x.shape)}, {tuple(weight.shape)}')
    return _P1.apply(x, weight, bias, float(alpha), float(bias_mul))
