"""Oracle (test infrastructure): pure-PyTorch CPU restatement of the reference's custom ops.

All functions are built from differentiable torch primitives, so first- and
second-order autograd come for free and serve as the reference for the HIP
ops' hand-written backward / double-backward kernels.
"""
import math

import torch
import torch.nn.functional as F


def upfirdn2d_out_size(in_size, up, down, pad0, pad1, ksize):
    """Output extent along one axis (reference: op/upfirdn2d.py:103-104,
    op/upfirdn2d_kernel.cu:237-240)."""
    return (in_size * up + pad0 + pad1 - ksize) // down + 1


def upfirdn2d_ref(x, kernel, up=1, down=1, pad=(0, 0)):
    """Upsample (zero-stuff) -> pad/crop -> FIR -> downsample.

    Restates the CPU path of the reference (op/upfirdn2d.py:159-200): the same
    `up`, `down` and `(pad0, pad1)` are used on both axes (op/upfirdn2d.py:145-156).
    x: [N, C, H, W]; kernel: [kh, kw].
    """
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    p0, p1 = pad
    t = x.reshape(n * c, 1, h, w)
    if up > 1:
        z = t.new_zeros(n * c, 1, h * up, w * up)
        z[:, :, ::up, ::up] = t
        t = z
    # positive pads add zeros, negative pads crop (op/upfirdn2d.py:171-180)
    t = F.pad(t, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    hh, ww = t.shape[2], t.shape[3]
    t = t[:, :, max(-p0, 0): hh - max(-p1, 0), max(-p0, 0): ww - max(-p1, 0)]
    # true convolution == correlation with the flipped kernel (op/upfirdn2d.py:187-188)
    wk = torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw).to(t.dtype)
    t = F.conv2d(t, wk)
    t = t[:, :, ::down, ::down]
    oh = upfirdn2d_out_size(h, up, down, p0, p1, kh)
    ow = upfirdn2d_out_size(w, up, down, p0, p1, kw)
    return t.reshape(n, c, oh, ow)


def fused_leaky_relu_ref(x, bias, negative_slope=0.2, scale=2 ** 0.5):
    """scale * leaky_relu(x + bias[c]) (reference: op/fused_bias_act_kernel.cu:28-47
    with act=3, grad=0; op/fused_act.py:51-70). bias may be None/empty."""
    if bias is not None and bias.numel() > 0:
        x = x + bias.reshape(1, -1, *([1] * (x.ndim - 2)))
    return F.leaky_relu(x, negative_slope) * scale


def fused_bias_act_ref(x, bias, ref, act, grad, alpha, scale):
    """The raw extension entry (op/fused_bias_act.cpp:11-21, kernel switch at
    op/fused_bias_act_kernel.cu:36-45).  Empty tensors mean 'absent'."""
    if bias is not None and bias.numel() > 0:
        x = x + bias.reshape(1, -1, *([1] * (x.ndim - 2)))
    if act == 1:
        y = x if grad < 2 else torch.zeros_like(x)
    elif act == 3:
        if grad == 0:
            y = torch.where(x > 0, x, x * alpha)
        elif grad == 1:
            y = torch.where(ref > 0, x, x * alpha)
        else:
            y = torch.zeros_like(x)
    else:
        y = x
    return y * scale


def make_blur_kernel(k):
    """Normalised separable FIR taps (model_probe_tune.py:29-37)."""
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = torch.outer(k, k)
    return k / k.sum()


def pixel_norm_ref(z):
    """model_probe_tune.py:25-26"""
    return z * torch.rsqrt(z.pow(2).mean(dim=1, keepdim=True) + 1e-8)


def equal_linear_ref(x, weight, bias, lr_mul=1.0, activation=False):
    """model_probe_tune.py:139-168"""
    scale = (1.0 / math.sqrt(weight.shape[1])) * lr_mul
    out = F.linear(x, weight * scale)
    if activation:
        return fused_leaky_relu_ref(out, bias * lr_mul)
    if bias is not None:
        out = out + bias * lr_mul
    return out


def modulated_conv2d_ref(x, style, weight, mod_w, mod_b, demodulate=True, upsample=False,
                         blur_kernel=None):
    """Per-sample modulated weights + grouped conv, exactly the reference's
    formulation (model_probe_tune.py:243-284).  weight: [1, Co, Ci, k, k]."""
    b, ci, h, w = x.shape
    _, co, _, k, _ = weight.shape
    s = equal_linear_ref(style, mod_w, mod_b).reshape(b, 1, ci, 1, 1)
    wgt = (1.0 / math.sqrt(ci * k * k)) * weight * s
    if demodulate:
        d = torch.rsqrt(wgt.pow(2).sum([2, 3, 4]) + 1e-8)
        wgt = wgt * d.reshape(b, co, 1, 1, 1)
    if upsample:
        xin = x.reshape(1, b * ci, h, w)
        wt = wgt.transpose(1, 2).reshape(b * ci, co, k, k)
        out = F.conv_transpose2d(xin, wt, padding=0, stride=2, groups=b)
        out = out.reshape(b, co, out.shape[2], out.shape[3])
        # Blur(pad=(1,1), kernel*4) for k=3, 4-tap filter (model_probe_tune.py:209-215)
        p = (blur_kernel.shape[0] - 2) - (k - 1)
        out = upfirdn2d_ref(out, blur_kernel * 4.0, pad=((p + 1) // 2 + 1, p // 2 + 1))
    else:
        xin = x.reshape(1, b * ci, h, w)
        out = F.conv2d(xin, wgt.reshape(b * co, ci, k, k), padding=k // 2, groups=b)
        out = out.reshape(b, co, out.shape[2], out.shape[3])
    return out


def minibatch_stddev_ref(x, stddev_group=25, stddev_feat=1):
    """model_probe_tune.py:748-756"""
    b, c, h, w = x.shape
    g = min(b, stddev_group)
    s = x.reshape(g, -1, stddev_feat, c // stddev_feat, h, w)
    s = torch.sqrt(s.var(0, unbiased=False) + 1e-8)
    s = s.mean([2, 3, 4], keepdim=True).squeeze(2)
    s = s.repeat(g, 1, h, w)
    return torch.cat([x, s], 1)
