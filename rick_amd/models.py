"""StyleGAN2 Generator / Discriminator on the MI355X engine, with the reference's constructor
and forward signatures and state_dict keys (gan_training/models/model_probe_tune.py:373-764;
key list: SURVEY.md §8a row M*), so rosinality checkpoints load and the RICK Fisher / mask code
can address parameters by the same names.

Differences that do not change results (documented in DESIGN.md):
  * activations are channels-last; RGB tensors are planar; the image returned by the
    Generator is made NCHW-contiguous;
  * modulated convolutions share one weight over the batch (modulation / demodulation folded
    into the MFMA kernel) instead of B grouped convolutions;
  * the Discriminator evaluates conv1 / conv2 of each ResBlock once and returns the same 14
    `feat` tensors (the reference computes them twice, model_probe_tune.py:740-744).
"""
import math
import os
import random

import torch
from torch import autograd, nn
from torch.nn import functional as F

from . import op
from .op import modconv as _mc
from .op.fused_act import FusedLeakyReLU, fused_leaky_relu, fused_noise_bias_act
from .op.upfirdn2d import upfirdn2d

CHANNELS = {4: 512, 8: 512, 16: 512, 32: 512}
# tests / tools switch the one-node split-image ResBlock (op/dblock.py) off to compare with the per-layer path
_USE_DBLOCK = not os.environ.get('RICK_NO_DBLOCK')
_USE_RGB_FORK = not os.environ.get('RICK_NO_RGB_FORK')
_USE_DEMOD_BANK = not os.environ.get('RICK_NO_DEMOD_BANK')      # (A/B switch, tools/ab_*.sh)
_USE_OWN_LINEAR = not os.environ.get('RICK_NO_OWN_LINEAR')      # EqualLinear with gradients on op.linear (else torch.addmm)


def _channels(res, channel_multiplier):
    return CHANNELS.get(res, (16384 // res) * channel_multiplier)   # 64:256cm 128:128cm 256:64cm 512:32cm 1024:16cm


def make_kernel(k):
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = torch.outer(k, k)
    return k / k.sum()


class PixelNorm(nn.Module):
    def forward(self, x):
        return x * torch.rsqrt(x.pow(2).mean(dim=1, keepdim=True) + 1e-8)


class _Fir(nn.Module):
    """FIR holder with one buffer named ``kernel`` (Blur / Upsample / Downsample of the
    reference, model_probe_tune.py:40-98)."""

    def __init__(self, taps, pad, up=1, down=1, gain=1.0):
        super().__init__()
        self.register_buffer('kernel', make_kernel(taps) * gain)
        self.pad, self.up, self.down = pad, up, down

    def forward(self, x):
        return upfirdn2d(x, self.kernel, up=self.up, down=self.down, pad=self.pad)


def Blur(kernel, pad, upsample_factor=1):
    return _Fir(kernel, pad, gain=float(upsample_factor ** 2) if upsample_factor > 1 else 1.0)


def Upsample(kernel, factor=2):
    p = len(kernel) - factor
    return _Fir(kernel, ((p + 1) // 2 + factor - 1, p // 2), up=factor, gain=float(factor ** 2))


def Downsample(kernel, factor=2):
    p = len(kernel) - factor
    return _Fir(kernel, ((p + 1) // 2, p // 2), down=factor)


class EqualLinear(nn.Module):
    """model_probe_tune.py:139-173.  Short batches (every use inside the training loop: the mapping network and D's final
    layers at <= 16 rows) run the library's own one-pass, fixed-order products (op.equal_linear without a graph, op.linear with
    gradients of any order); longer batches and N-D inputs go to the BLAS library.  The bias + LeakyReLU tail is the fused op."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def _no_grad_needed(self, x):
        return not (torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad
                                                 or (self.bias is not None and self.bias.requires_grad)))

    def forward(self, x, pixelnorm=False):
        if (x.ndim == 2 and x.is_cuda and x.dtype == torch.float32 and x.shape[0] <= 16 and x.shape[1] % 4 == 0
                and self.activation in (None, 'fused_lrelu') and self._no_grad_needed(x)):
            # short-batch forward without autograd (the mapping network of every train step): one launch for
            # [PixelNorm +] GEMM + bias * lr_mul + activation instead of 3 (+ 5)
            return op.equal_linear(x, self.weight, self.bias, self.scale, self.lr_mul, bool(self.activation), pixelnorm)
        if pixelnorm:
            x = x * torch.rsqrt(torch.mean(x ** 2, dim=1, keepdim=True) + 1e-8)
        bias = self.bias if self.lr_mul == 1 or self.bias is None else self.bias * self.lr_mul
        if _USE_OWN_LINEAR and op._linear.supported(x, self.weight):
            # with gradients, short batch (D's final layers in every D pass): forward, data and weight gradient are one pass
            # over the matrix each, summed in a fixed order (rick_amd/csrc/linear.hip)
            if self.activation:
                return fused_leaky_relu(op.linear(x, self.weight, None, self.scale), bias)
            return op.linear(x, self.weight, self.bias, self.scale, self.lr_mul)
        # x @ (W * scale)^T as one rocBLAS call (scale = GEMM alpha) instead of a [out, in] elementwise pass
        if x.ndim != 2:
            out = F.linear(x, self.weight * self.scale)
            if self.activation:
                return fused_leaky_relu(out, bias)
            return out if bias is None else out + bias
        if self.activation or bias is None:
            out = torch.addmm(x.new_empty(1), x, self.weight.t(), beta=0, alpha=self.scale)
            return fused_leaky_relu(out, bias) if self.activation else out
        return torch.addmm(bias, x, self.weight.t(), beta=1, alpha=self.scale)


class EqualConv2d(nn.Module):
    """model_probe_tune.py:101-136 on the MFMA conv kernels (weight scale folded into packing)."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride, self.padding = stride, padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, x):
        O, I, k, _ = self.weight.shape
        if I <= 4 and k == 1 and self.stride == 1:
            # image -> features (discriminator input, model_probe_tune.py:679): thin product
            W = (self.weight.view(O, I) * self.scale).t().unsqueeze(0)          # [1, J=I, C=O]
            y = op.thin_bwdx(x.contiguous(), W)
        else:
            y = op.conv2d(x, self.weight, self.stride, self.padding, wscale=self.scale, key=(self.weight, 'w'))
        if self.bias is not None:
            y = y + self.bias.view(1, -1, 1, 1)
        return y


class ModulatedConv2d(nn.Module):
    """model_probe_tune.py:188-284 (3x3 plain / 3x3 upsample / 1x1 no-demod)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.eps = 1e-8
        self.kernel_size, self.in_channel, self.out_channel = kernel_size, in_channel, out_channel
        self.upsample, self.downsample = upsample, downsample
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1), upsample_factor=factor)
        if downsample:      # model_probe_tune.py:214-220 (no network of the reference uses it)
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2, p // 2))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate

    def forward(self, x, style, tail=None, s=None, d=None, rgb_tail=None, next_s=None, fork=False):
        """tail = (bias, noise, noise_weight, negative_slope, gain): apply StyledConv's NoiseInjection + FusedLeakyReLU
        as part of this layer (fused into the blur launch of the upsampling variant; first-order mode only).
        s = modulation(style) (and d, the demodulation coefficients) when the Generator has already evaluated them for
        all layers at once."""
        if s is None:
            s = self.modulation(style)                               # [B, Ci]
        # [Co, Ci, k, k] as a VIEW (not weight[0]: a select's backward materialises zeros + a copy of the whole weight per use
        # whenever the gradient travels through autograd — 41 of them in a path-length step)
        w = self.weight.view(self.out_channel, self.in_channel, self.kernel_size, self.kernel_size)
        if self.kernel_size == 1 and self.out_channel <= 4 and not self.demodulate:
            if rgb_tail is not None and not op.second_order_enabled():
                # ToRGB: weight modulation, bias and the skip addition inside the one launch (first-order steps)
                f = op.torgb_fork if fork else op.torgb     # fork: -> (x for the next layer, rgb), one node for the branch point
                return f(x, w.view(self.out_channel, self.in_channel), s, rgb_tail[0], rgb_tail[1], self.scale)
            Wn = (self.scale * w.view(1, self.out_channel, self.in_channel)) * s.unsqueeze(1)   # [B, 3, Ci]
            out = op.thin_fwd(x, Wn)                                 # planar [B, 3, H, W]
            if rgb_tail is not None:
                out = out + rgb_tail[0]
                out = out + rgb_tail[1] if rgb_tail[1] is not None else out
            return out
        key = (self.weight, 'mod')
        if self.downsample:
            # blur -> stride-2 modulated conv (model_probe_tune.py:270-276), composed from the twice-differentiable
            # primitives: y = d * conv_s2(W, blur(s * x)); the blur is per channel, so it commutes with the scale s
            dd = _mc.demod_coeff(w, s, self.scale, self.eps) if self.demodulate else None
            y = op.conv2d(self.blur(op.chan_scale(x, s)), w, 2, 0, wscale=self.scale, key=key)
            return op.chan_scale(y, dd) if dd is not None else y
        if self.demodulate and d is None:     # composed tensor algebra when a second derivative is needed, 2 + 2 launches otherwise
            d = (_mc.demod_coeff(w, s, self.scale, self.eps) if op.second_order_enabled()
                 else _mc.demod_coeff_fused(w, s, self.scale, self.eps, key))
        if op.second_order_enabled():
            y = _mc.modulated_conv_composed(x, w, s, None, self.scale, self.upsample, key)
            if self.upsample:
                y = self.blur(y)
            return op.chan_scale(y, d) if d is not None else y
        if tail is not None and not self.upsample and self.out_channel % 4 == 0:
            return _mc.modulated_conv_fused(x, w, s, d, self.scale, False, key, tail, next_s=next_s)     # tail in the conv epilogue
        y = _mc.modulated_conv_fused(x, w, s, d, self.scale, self.upsample, key)
        if tail is not None and self.upsample:
            # tail in the blur launch; next_s / d: split-image hand-over to the next layer's forward and to this layer's backward
            return op.upfirdn2d_noise_bias_act(y, self.blur.kernel, self.blur.pad, *tail, next_s=next_s,
                                               bwd_scale=d if _mc._USE_SPLIT else None)
        y = self.blur(y) if self.upsample else y
        return y if tail is None else fused_noise_bias_act(y, *tail)


class NoiseInjection(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise=None):      # standalone use; StyledConv fuses it with the activation
        if noise is None:
            b, _, h, w = image.shape
            noise = image.new_empty(b, 1, h, w).normal_()
        return image + self.weight * noise


class ConstantInput(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, x):
        return self.input.expand(x.shape[0], -1, -1, -1).contiguous(memory_format=torch.channels_last)


class StyledConv(nn.Module):
    """conv -> noise -> bias+LeakyReLU (model_probe_tune.py:314-348); the last two are one kernel."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1],
                 demodulate=True):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate)
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)

    def forward(self, x, style, noise=None, s=None, d=None, next_s=None):
        if noise is None:
            r = x.shape[2] * 2 if self.conv.upsample else x.shape[2]
            noise = torch.empty(x.shape[0], 1, r, r * x.shape[3] // x.shape[2], device=x.device, dtype=x.dtype).normal_()
        tail = (self.activate.bias, noise, self.noise.weight, self.activate.negative_slope, self.activate.scale)
        if op.second_order_enabled():
            return fused_noise_bias_act(self.conv(x, style, s=s, d=d), *tail)
        return self.conv(x, style, tail, s=s, d=d, next_s=next_s)


class ToRGB(nn.Module):
    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def forward(self, x, style, skip=None, s=None, fork=False):
        # out = conv(x, style) + bias; out = out + upsample(skip)   (model_probe_tune.py:366-370)
        # fork (first-order CUDA path): returns (x, out) — x to be used by the next layer, see op.torgb_fork
        return self.conv(x, style, s=s, rgb_tail=(self.bias, self.upsample(skip) if skip is not None else None), fork=fork)


class _Mapping(nn.Sequential):
    """The mapping network ``Generator.style`` = Sequential(PixelNorm, EqualLinear x n_mlp) (model_probe_tune.py:418-428,
    same child indices / state_dict keys); the PixelNorm is evaluated inside the first linear's launch."""

    def forward(self, x):
        mods = list(self)
        if len(mods) > 1 and isinstance(mods[0], PixelNorm) and isinstance(mods[1], EqualLinear) and x.ndim == 2:
            x = mods[1](x, pixelnorm=True)
            mods = mods[2:]
        for m in mods:
            x = m(x)
        return x


class _FisherMixin:
    def estimate_fisher(self, loglikelihood):
        """(grads, {name: grad**2}) for every parameter that requires grad
        (model_probe_tune.py:481-504, :706-729)."""
        named = [(n, p) for n, p in self.named_parameters()]
        grads = autograd.grad(loglikelihood, [p for _, p in named], retain_graph=True, allow_unused=True)
        est = {}
        for (n, p), g in zip(named, grads):
            if p.requires_grad:
                est[n] = g.detach() ** 2 if g is not None else p.detach().clone().zero_()
        return grads, est


class Generator(nn.Module, _FisherMixin):
    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01):
        super().__init__()
        self.size, self.style_dim = size, style_dim
        self.style = _Mapping(PixelNorm(), *[EqualLinear(style_dim, style_dim, lr_mul=lr_mlp,
                                                         activation='fused_lrelu') for _ in range(n_mlp)])
        self.channels = {r: _channels(r, channel_multiplier) for r in (4, 8, 16, 32, 64, 128, 256, 512, 1024)}
        self.input = ConstantInput(self.channels[4])
        self.conv1 = StyledConv(self.channels[4], self.channels[4], 3, style_dim, blur_kernel=blur_kernel)
        self.to_rgb1 = ToRGB(self.channels[4], style_dim, upsample=False)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.convs, self.upsamples, self.to_rgbs = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.noises = nn.Module()
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer(f'noise_{layer_idx}', torch.randn(1, 1, 2 ** res, 2 ** res))
        in_channel = self.channels[4]
        for i in range(3, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            self.convs.append(StyledConv(in_channel, out_channel, 3, style_dim, upsample=True, blur_kernel=blur_kernel))
            self.convs.append(StyledConv(out_channel, out_channel, 3, style_dim, blur_kernel=blur_kernel))
            self.to_rgbs.append(ToRGB(out_channel, style_dim))
            in_channel = out_channel
        self.n_latent = self.log_size * 2 - 2

    def _modulation_bank(self):
        """ModulationBank over the modulation EqualLinears in forward order (conv1, to_rgb1, then convs / convs / to_rgb per
        resolution) with the latent row each of them reads."""
        bank = self.__dict__.get('_modbank')
        if bank is None:
            mods, idx = [self.conv1.conv.modulation, self.to_rgb1.conv.modulation], [0, 1]
            i = 1
            for blk, to_rgb in enumerate(self.to_rgbs):
                mods += [self.convs[2 * blk].conv.modulation, self.convs[2 * blk + 1].conv.modulation, to_rgb.conv.modulation]
                idx += [i, i + 1, i + 2]
                i += 2
            bank = self.__dict__['_modbank'] = _mc.ModulationBank(mods, idx)
        return bank

    def _demod_bank(self):
        """DemodBank over the demodulated convolutions in forward order (conv1, then the two StyledConvs of every resolution)
        with the position of each one's style vector in the modulation bank's output."""
        bank = self.__dict__.get('_dmbank')
        if bank is None:
            convs, idx = [self.conv1.conv], [0]
            for blk in range(len(self.to_rgbs)):
                convs += [self.convs[2 * blk].conv, self.convs[2 * blk + 1].conv]
                idx += [2 + 3 * blk, 3 + 3 * blk]
            bank = self.__dict__['_dmbank'] = _mc.DemodBank(convs, idx, self._modulation_bank().C)
        return bank

    def _styles_batched(self, latent):
        """Path-length step (second-order autograd): the per-layer style algebra — s = EqualLinear(latent row),
        d = rsqrt(s^2 @ (scale^2 sum_k W^2)^T + eps), model_probe_tune.py:246-252 — is [B, 512]-sized tensor work that
        costs ~7 launches per layer forward and several dozen in each of the two backward passes (measured: ~700 of the
        ~1 950 launches of a path-length step).  Layers of equal shape are stacked and evaluated by batched tensor ops
        (plain torch: differentiable to any order), so the count no longer scales with the depth of the network."""
        bank = self._modulation_bank()
        convs = [self.conv1.conv, self.to_rgb1.conv]
        for blk, to_rgb in enumerate(self.to_rgbs):
            convs += [self.convs[2 * blk].conv, self.convs[2 * blk + 1].conv, to_rgb.conv]
        n = len(convs)
        s_out, d_out = [None] * n, [None] * n
        groups = {}
        for l, m in enumerate(bank.linears):
            groups.setdefault(m.weight.shape[0], []).append(l)
        for C, ls in groups.items():
            W = torch.stack([bank.linears[l].weight for l in ls])                       # [L, C, K]
            b = torch.stack([bank.linears[l].bias for l in ls])                         # [L, C]
            cache = self.__dict__.setdefault('_lat_index', {})                          # device index tensors (built on the first,
            key = (C, latent.device)                                                    # eager, call: no H2D copy under capture)
            if key not in cache:
                cache[key] = torch.tensor([bank.lat_idx[l] for l in ls], device=latent.device)
            L = latent.index_select(1, cache[key]).transpose(0, 1)                      # [L, B, K]
            S = torch.baddbmm(b.unsqueeze(1), L, W.transpose(1, 2), alpha=bank.scale)   # [L, B, C]
            for l, sl in zip(ls, S.unbind(0)):      # one unbind: its backward is a single stack
                s_out[l] = sl
        groups = {}
        for l, c in enumerate(convs):
            if c.demodulate:
                groups.setdefault(tuple(c.weight.shape), []).append(l)
        for shape, ls in groups.items():
            Wc = torch.stack([convs[l].weight.squeeze(0) for l in ls])                  # [L, O, I, k, k]
            wsq = (Wc * convs[ls[0]].scale).pow(2).sum([3, 4])                          # [L, O, I]
            S2 = torch.stack([s_out[l] for l in ls]).pow(2)                             # [L, B, I]
            D = torch.rsqrt(torch.bmm(S2, wsq.transpose(1, 2)) + convs[ls[0]].eps)      # [L, B, O]
            for l, dl in zip(ls, D.unbind(0)):
                d_out[l] = dl
        return s_out, d_out

    def make_noise(self):
        device = self.input.input.device
        noises = [torch.randn(1, 1, 4, 4, device=device)]
        for i in range(3, self.log_size + 1):
            noises += [torch.randn(1, 1, 2 ** i, 2 ** i, device=device) for _ in range(2)]
        return noises

    def mean_latent(self, n_latent):
        z = torch.randn(n_latent, self.style_dim, device=self.input.input.device)
        return self.style(z).mean(0, keepdim=True)

    def get_latent(self, input):
        return self.style(input)

    def forward(self, styles, return_latents=False, inject_index=None, truncation=1, truncation_latent=None,
                input_is_latent=False, noise=None, randomize_noise=True, return_feats=False):
        if not input_is_latent:
            styles = [self.style(s) for s in styles]
        if noise is None:
            noise = ([None] * self.num_layers if randomize_noise
                     else [getattr(self.noises, f'noise_{i}') for i in range(self.num_layers)])
        if truncation < 1:
            styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
        if len(styles) < 2:
            inject_index = self.n_latent
            latent = styles[0].unsqueeze(1).repeat(1, inject_index, 1) if styles[0].ndim < 3 else styles[0]
        else:
            if inject_index is None:
                inject_index = random.randint(1, self.n_latent - 1)
            latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                                styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)], 1)
        if randomize_noise and latent.is_cuda and all(n is None for n in noise):
            # NoiseInjection draws image.new_empty(b, 1, h, w).normal_() per layer (model_probe_tune.py:294-296): the
            # same i.i.d. N(0, 1) maps from ONE generator launch (per-layer views of one buffer)
            B = latent.shape[0]
            res = [4] + [2 ** i for i in range(3, self.log_size + 1) for _ in range(2)]
            flat = torch.empty(B * sum(r * r for r in res), device=latent.device, dtype=latent.dtype).normal_()
            noise, o = [], 0
            for r in res:
                noise.append(flat[o:o + B * r * r].view(B, 1, r, r))
                o += B * r * r
        feats = []
        # one unbind instead of 20 selects: its backward is a single stack, not zeros + slice-copy + add per use
        lat = latent.unbind(1)
        # every modulation linear of the network in one launch (first-order steps whose latent needs no gradient: the D / G
        # train steps and inference; the path-length step and the Fisher sweep take the per-layer, twice-differentiable path)
        sb = [None] * (2 + 3 * len(self.to_rgbs))
        db = list(sb)
        if (latent.is_cuda and not op.second_order_enabled() and not latent.requires_grad and latent.ndim == 3
                and latent.shape[0] <= 8 and latent.shape[1] == self.n_latent and latent.dtype == torch.float32):
            sb = self._modulation_bank()(latent)
            if _USE_DEMOD_BANK:     # every layer's demodulation coefficients from one launch (wsq cached per weight update)
                dl = self._demod_bank()(sb)
                if dl is not None:
                    bank = self._demod_bank()
                    for j, d_l in zip(bank.s_index, dl):
                        db[j] = d_l
        elif latent.is_cuda and op.second_order_enabled() and latent.ndim == 3 and latent.shape[1] == self.n_latent:
            sb, db = self._styles_batched(latent)
        # (next_s: the style scales of the following modulated convolution — a layer folds them into the split image it writes
        # for that convolution, op/modconv.py; first-order steps only, sb is None-filled otherwise)
        nblk = len(self.to_rgbs)
        fork = (_USE_RGB_FORK and latent.is_cuda and latent.dtype == torch.float32 and not op.second_order_enabled()
                and torch.is_grad_enabled() and not return_feats)
        out = self.conv1(self.input(latent), lat[0], noise=noise[0], s=sb[0], d=db[0], next_s=sb[2] if nblk else None)
        feats.append(out)
        if fork and nblk:
            out, skip = self.to_rgb1(out, lat[1], s=sb[1], fork=True)
        else:
            skip = self.to_rgb1(out, lat[1], s=sb[1])
        i = 1
        for blk, to_rgb in enumerate(self.to_rgbs):
            out = self.convs[2 * blk](out, lat[i], noise=noise[2 * blk + 1], s=sb[2 + 3 * blk], d=db[2 + 3 * blk],
                                      next_s=sb[3 + 3 * blk])
            feats.append(out)
            out = self.convs[2 * blk + 1](out, lat[i + 1], noise=noise[2 * blk + 2], s=sb[3 + 3 * blk], d=db[3 + 3 * blk],
                                          next_s=sb[5 + 3 * blk] if blk + 1 < nblk else None)
            feats.append(out)
            if fork and blk + 1 < nblk:      # `out` also feeds the next resolution: both gradients meet in ONE node (op.torgb_fork)
                out, skip = to_rgb(out, lat[i + 2], skip, s=sb[4 + 3 * blk], fork=True)
            else:
                skip = to_rgb(out, lat[i + 2], skip, s=sb[4 + 3 * blk])
            i += 2
        image = skip.contiguous()
        if return_latents:
            return image, latent
        if return_feats:
            return image, feats
        return image, None


class ScaledLeakyReLU(nn.Module):
    """model_probe_tune.py:176-185: sqrt(2) * leaky_relu(x, 0.2) — the fused bias + activation kernel without a bias."""

    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, x):
        return fused_leaky_relu(x, None, self.negative_slope, math.sqrt(2))


class ConvLayer(nn.Sequential):
    """[Blur] + EqualConv2d + [FusedLeakyReLU] with the reference's child indices
    (model_probe_tune.py:595-641) — the Fisher code derives bias keys from them."""

    def __init__(self, in_channel, out_channel, kernel_size, downsample=False, blur_kernel=[1, 3, 3, 1], bias=True,
                 activate=True):
        layers = []
        if downsample:
            p = (len(blur_kernel) - 2) + (kernel_size - 1)
            layers.append(Blur(blur_kernel, pad=((p + 1) // 2, p // 2)))
            stride, self.padding = 2, 0
        else:
            stride, self.padding = 1, kernel_size // 2
        layers.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=self.padding, stride=stride,
                                  bias=bias and not activate))
        if activate:
            layers.append(FusedLeakyReLU(out_channel) if bias else ScaledLeakyReLU(0.2))
        super().__init__(*layers)
        self._fusable = activate and bias and out_channel % 4 == 0 and not (in_channel <= 4 and kernel_size == 1)

    def forward(self, x):
        # conv -> bias + LeakyReLU as ONE launch (tail in the MFMA kernel's epilogue) when only first derivatives are
        # needed; the child modules (and their state_dict keys) stay exactly the reference's
        if (_USE_DBLOCK and not self._fusable and len(self) == 2 and isinstance(self[1], FusedLeakyReLU) and x.is_cuda
                and x.dtype == torch.float32 and not op.second_order_enabled() and op.get_precision() == 'fp16x3'):
            conv, act = self[0], self[1]
            O, I, k, _ = conv.weight.shape
            if I <= 4 and k == 1 and conv.stride == 1 and conv.bias is None and O % 4 == 0:
                # image -> features (the discriminator's input layer): one launch, fp32 + the split image the first ResBlock reads
                from .op import dblock
                return dblock.d_input(x, conv.weight, act.bias, conv.scale, act.negative_slope, act.scale,
                                      lambda x_, w_, b_: fused_leaky_relu(
                                          op.thin_bwdx(x_.contiguous(), (w_.view(O, I) * conv.scale).t().unsqueeze(0)), b_,
                                          act.negative_slope, act.scale))
        if not self._fusable or op.second_order_enabled():
            return super().forward(x)
        mods = list(self)
        for m in mods[:-2]:
            x = m(x)
        conv, act = mods[-2], mods[-1]
        return op.conv2d_bias_act(x, conv.weight, act.bias, conv.stride, conv.padding, wscale=conv.scale,
                                  key=(conv.weight, 'w'), negative_slope=act.negative_slope, gain=act.scale)


class ResBlock(nn.Module):
    def __init__(self, in_channel, out_channel, blur_kernel=[1, 3, 3, 1], downsample=True):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=downsample)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=downsample, activate=False, bias=False)

    def _skip(self, x):
        """skip = 1x1 stride-2 conv of blur(x) (model_probe_tune.py:650-652).  A stride-2 1x1 conv only reads
        the even positions of the blurred map, so the FIR is evaluated directly at those positions
        (upfirdn2d down=2: same taps, same order per output -> identical values, 4x fewer of them) and the
        1x1 conv runs dense at stride 1."""
        blur, conv = self.skip[0], self.skip[1]
        if not isinstance(blur, _Fir) or conv.stride != 2 or conv.weight.shape[2] != 1:
            return self.skip(x)
        y = upfirdn2d(x, blur.kernel, up=1, down=2, pad=blur.pad)
        return op.conv2d(y, conv.weight, 1, 0, wscale=conv.scale, key=(conv.weight, 'w'))

    def _per_layer(self, x):
        t1 = self.conv1(x)
        t2 = self.conv2(t1)
        return op.add_scale(t2, self._skip(x), 1 / math.sqrt(2)), t1, t2

    def _fused_cfg(self):
        """Arguments of the one-node, split-image form of the block (op/dblock.py), or None when the block is not the
        reference's standard shape (3x3 -> blur + 3x3 stride 2, blur + 1x1 stride 2 skip, one shared 4x4 FIR).  Built per
        call from the live modules (a cached tuple would keep pointing at the parameters of a module this one was copied
        from); only the shape test is cached."""
        c1, c2, sk = list(self.conv1), list(self.conv2), list(self.skip)
        std = self.__dict__.get('_dblock_std')
        if std is None:
            std = (len(c1) == 2 and len(c2) == 3 and len(sk) == 2 and isinstance(c2[0], _Fir) and isinstance(sk[0], _Fir)
                   and isinstance(c1[1], FusedLeakyReLU) and isinstance(c2[2], FusedLeakyReLU) and c2[1].stride == 2
                   and sk[1].stride == 2 and sk[1].bias is None and c1[0].bias is None and c2[1].bias is None
                   and tuple(c2[0].kernel.shape) == (4, 4) and c2[0].up == c2[0].down == 1 and sk[0].up == sk[0].down == 1
                   and c1[1].negative_slope == c2[2].negative_slope and c1[1].scale == c2[2].scale
                   and bool(torch.equal(c2[0].kernel, sk[0].kernel)) and bool((c2[0].kernel >= 0).all())
                   and abs(float(c2[0].kernel.sum()) - 1.0) < 1e-5)      # (the blur's bound is its input's maximum)
            self.__dict__['_dblock_std'] = std
        if not std:
            return None

        def compose(x_, w1_, b1_, w2_, b2_, ws_):      # the same block from the twice-differentiable per-layer ops
            t1 = op.conv2d_bias_act(x_, w1_, b1_, 1, 1, wscale=c1[0].scale, key=(c1[0].weight, 'w'),
                                    negative_slope=c1[1].negative_slope, gain=c1[1].scale)
            t2 = op.conv2d_bias_act(c2[0](t1), w2_, b2_, 2, 0, wscale=c2[1].scale, key=(c2[1].weight, 'w'),
                                    negative_slope=c2[2].negative_slope, gain=c2[2].scale)
            y = upfirdn2d(x_, sk[0].kernel, up=1, down=2, pad=sk[0].pad)
            s_ = op.conv2d(y, ws_, 1, 0, wscale=sk[1].scale, key=(sk[1].weight, 'w'))
            return op.add_scale(t2, s_, 1 / math.sqrt(2)), t1, t2
        return (c1[0].scale, c2[1].scale, sk[1].scale, c1[1].negative_slope, c1[1].scale, c2[0].pad, sk[0].pad,
                c1[0].weight, c2[1].weight, sk[1].weight, compose)

    def forward(self, x, feat=None):
        from .op import dblock
        cfg = None
        if (x.is_cuda and x.dtype == torch.float32 and not op.second_order_enabled() and op.get_precision() == 'fp16x3' and _USE_DBLOCK
                and dblock.block_supported(x, self.conv1[0].weight, self.conv2[-2].weight, self.skip[-1].weight)):
            cfg = self._fused_cfg()
        if cfg is not None:
            out, t1, t2 = dblock.d_resblock(x, self.conv1[0].weight, self.conv1[1].bias, self.conv2[1].weight, self.conv2[2].bias,
                                            self.skip[1].weight, self.conv2[0].kernel, cfg)
        else:
            out, t1, t2 = self._per_layer(x)
        if feat is not None:
            feat += [t1, t2]
        return out


class Discriminator(nn.Module, _FisherMixin):
    def __init__(self, size, channel_multiplier=2, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        ch = {r: _channels(r, channel_multiplier) for r in (4, 8, 16, 32, 64, 128, 256, 512, 1024)}
        convs = [ConvLayer(3, ch[size], 1)]
        log_size = int(math.log(size, 2))
        in_channel = ch[size]
        for i in range(log_size, 2, -1):
            out_channel = ch[2 ** (i - 1)]
            convs.append(ResBlock(in_channel, out_channel, blur_kernel))
            in_channel = out_channel
        self.convs = nn.Sequential(*convs)
        self.stddev_group, self.stddev_feat = 25, 1
        self.final_conv = ConvLayer(in_channel + 1, ch[4], 3)
        self.final_linear = nn.Sequential(EqualLinear(ch[4] * 4 * 4, ch[4], activation='fused_lrelu'),
                                          EqualLinear(ch[4], 1))

    def forward(self, inp, ind=None, real=False, calls=1):
        """Reference signature (model_probe_tune.py:732) plus `calls`: the batch is the concatenation of
        `calls` independent calls (minibatch-stddev statistics stay per call), e.g. D(cat(fake, real), calls=2)
        returns exactly cat(D(fake), D(real))."""
        feat = []
        x = self.convs[0](inp)
        feat.append(x)
        for blk in list(self.convs)[1:]:
            x = blk(x, feat)
        out = op.minibatch_stddev(x, self.stddev_group, self.stddev_feat, second_order=op.second_order_enabled(),
                                  calls=calls)
        out = self.final_conv(out)
        feat.append(out)
        out = out.contiguous().view(out.shape[0], -1)      # NCHW flatten order, as the checkpoint expects
        return self.final_linear(out), feat
