"""Oracle (test infrastructure): functional CPU restatement of the reference
Generator / Discriminator (gan_training/models/model_probe_tune.py), driven by
a plain ``{key: tensor}`` state dict with the reference's key names
(SURVEY.md §8a row M*).  Pure PyTorch; differentiable to any order.
"""
import math
import random

import torch
import torch.nn.functional as F

from .ops_ref import (equal_linear_ref, fused_leaky_relu_ref, make_blur_kernel,
                      minibatch_stddev_ref, modulated_conv2d_ref, pixel_norm_ref,
                      upfirdn2d_ref)

BLUR = (1, 3, 3, 1)


def channels_for(size_px, channel_multiplier=2):
    """model_probe_tune.py:400-410 / :667-677"""
    return {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier,
            128: 128 * channel_multiplier, 256: 64 * channel_multiplier,
            512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}[size_px]


# ----------------------------------------------------------------------------- generator

def mapping_ref(sd, z, n_mlp=8, lr_mlp=0.01):
    """PixelNorm + n_mlp x EqualLinear(lr_mul, fused lrelu) (model_probe_tune.py:389-398)."""
    w = pixel_norm_ref(z)
    for i in range(1, n_mlp + 1):
        w = equal_linear_ref(w, sd[f'style.{i}.weight'], sd[f'style.{i}.bias'],
                             lr_mul=lr_mlp, activation=True)
    return w


def _styled_conv(sd, prefix, x, w_lat, noise, upsample, blur_k):
    """StyledConv = ModulatedConv2d + NoiseInjection + FusedLeakyReLU (model_probe_tune.py:314-348)."""
    out = modulated_conv2d_ref(x, w_lat, sd[f'{prefix}.conv.weight'],
                               sd[f'{prefix}.conv.modulation.weight'],
                               sd[f'{prefix}.conv.modulation.bias'],
                               demodulate=True, upsample=upsample, blur_kernel=blur_k)
    if noise is None:
        noise = torch.randn(out.shape[0], 1, out.shape[2], out.shape[3], dtype=out.dtype)
    out = out + sd[f'{prefix}.noise.weight'] * noise
    return fused_leaky_relu_ref(out, sd[f'{prefix}.activate.bias'])


def _to_rgb(sd, prefix, x, w_lat, skip, blur_k):
    """ToRGB: 1x1 modulated conv without demod + bias + upsampled skip (model_probe_tune.py:351-370)."""
    out = modulated_conv2d_ref(x, w_lat, sd[f'{prefix}.conv.weight'],
                               sd[f'{prefix}.conv.modulation.weight'],
                               sd[f'{prefix}.conv.modulation.bias'], demodulate=False)
    out = out + sd[f'{prefix}.bias']
    if skip is not None:
        # Upsample(factor 2): kernel*4, pad=(2,1) for a 4-tap filter (model_probe_tune.py:40-58)
        out = out + upfirdn2d_ref(skip, blur_k * 4.0, up=2, down=1, pad=(2, 1))
    return out


def generator_ref(sd, styles, size=256, n_mlp=8, return_latents=False, inject_index=None,
                  truncation=1, truncation_latent=None, input_is_latent=False, noise=None,
                  randomize_noise=True, return_feats=False):
    """Generator.forward (model_probe_tune.py:509-592). `styles` is a list of [B,512]."""
    log_size = int(math.log2(size))
    num_layers = (log_size - 2) * 2 + 1
    n_latent = log_size * 2 - 2
    blur_k = make_blur_kernel(BLUR).to(styles[0].dtype)
    if not input_is_latent:
        styles = [mapping_ref(sd, s, n_mlp) for s in styles]
    if noise is None:
        noise = ([None] * num_layers if randomize_noise
                 else [sd[f'noises.noise_{i}'] for i in range(num_layers)])
    if truncation < 1:
        styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
    if len(styles) < 2:
        latent = styles[0].unsqueeze(1).repeat(1, n_latent, 1) if styles[0].ndim < 3 else styles[0]
    else:
        if inject_index is None:
            inject_index = random.randint(1, n_latent - 1)
        latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                            styles[1].unsqueeze(1).repeat(1, n_latent - inject_index, 1)], 1)
    b = latent.shape[0]
    feats = []
    out = sd['input.input'].repeat(b, 1, 1, 1)
    out = _styled_conv(sd, 'conv1', out, latent[:, 0], noise[0], False, blur_k)
    feats.append(out)
    skip = _to_rgb(sd, 'to_rgb1', out, latent[:, 1], None, blur_k)
    i = 1
    for blk in range(log_size - 2):
        out = _styled_conv(sd, f'convs.{2 * blk}', out, latent[:, i], noise[2 * blk + 1], True, blur_k)
        feats.append(out)
        out = _styled_conv(sd, f'convs.{2 * blk + 1}', out, latent[:, i + 1], noise[2 * blk + 2], False, blur_k)
        feats.append(out)
        skip = _to_rgb(sd, f'to_rgbs.{blk}', out, latent[:, i + 2], skip, blur_k)
        i += 2
    if return_latents:
        return skip, latent
    if return_feats:
        return skip, feats
    return skip, None


# ------------------------------------------------------------------------- discriminator

def _equal_conv(x, weight, stride=1, padding=0):
    """EqualConv2d without bias (model_probe_tune.py:101-130)."""
    scale = 1.0 / math.sqrt(weight.shape[1] * weight.shape[2] * weight.shape[3])
    return F.conv2d(x, weight * scale, stride=stride, padding=padding)


def discriminator_ref(sd, img, size=256, stddev_group=25):
    """Discriminator.forward (model_probe_tune.py:732-764).  Returns (logit, feat) where feat
    holds the same 14 tensors; conv1/conv2 of each ResBlock are evaluated once (the reference
    evaluates them twice with identical values)."""
    log_size = int(math.log2(size))
    blur_k = make_blur_kernel(BLUR).to(img.dtype)
    feat = []
    x = fused_leaky_relu_ref(_equal_conv(img, sd['convs.0.0.weight']), sd['convs.0.1.bias'])
    feat.append(x)
    for blk in range(1, log_size - 1):
        p = f'convs.{blk}'
        t1 = fused_leaky_relu_ref(_equal_conv(x, sd[f'{p}.conv1.0.weight'], padding=1),
                                  sd[f'{p}.conv1.1.bias'])
        feat.append(t1)
        t2 = upfirdn2d_ref(t1, blur_k, pad=(2, 2))                     # ConvLayer(downsample) blur, k=3
        t2 = fused_leaky_relu_ref(_equal_conv(t2, sd[f'{p}.conv2.1.weight'], stride=2),
                                  sd[f'{p}.conv2.2.bias'])
        feat.append(t2)
        sk = upfirdn2d_ref(x, blur_k, pad=(1, 1))                       # skip blur, k=1
        sk = _equal_conv(sk, sd[f'{p}.skip.1.weight'], stride=2)
        x = (t2 + sk) / math.sqrt(2)
    out = minibatch_stddev_ref(x, stddev_group)
    out = fused_leaky_relu_ref(_equal_conv(out, sd['final_conv.0.weight'], padding=1),
                               sd['final_conv.1.bias'])
    feat.append(out)
    out = out.reshape(out.shape[0], -1)
    out = equal_linear_ref(out, sd['final_linear.0.weight'], sd['final_linear.0.bias'], activation=True)
    out = equal_linear_ref(out, sd['final_linear.1.weight'], sd['final_linear.1.bias'])
    return out, feat
