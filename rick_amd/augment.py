"""Adaptive discriminator augmentation: the second caller of ``op.upfirdn2d``.

Mirrors ``augment(img, p, transform_matrix=(None, None))`` of the reference (non_leaking.py:394-398):
a random 2-D affine map applied with 2x supersampling — reflect pad, 12x12 sym6 upsampling FIR
(``upfirdn2d(up=2)``), bilinear ``grid_sample``, 12x12 FIR + decimation (``upfirdn2d(down=2)``), crop —
followed by a random 4x4 colour transform.  The two FIR passes run on the generic HIP upfirdn2d kernel
(planar path, bit-exact index math); padding, grid sampling and the 3x3 colour product are library ops
on the device.  The transform matrices are sampled on the host exactly like the reference does (same
distributions, same draw order from torch's CPU generator, so a seed reproduces the reference's G / C);
they can also be passed in, which is how the parity tests pin the image path to the reference.
"""
import math

import torch
import torch.nn.functional as F

from .op import upfirdn2d

# sym6 decomposition low-pass filter (the reference's antialiasing kernel, non_leaking.py:9-22)
SYM6 = (0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633,
        0.4910559419267466, 0.787641141030194, 0.3379294217276218, -0.07263752278646252,
        -0.021060292512300564, 0.04472490177066578, 0.0017677118642428036, -0.007800708325034148)


# ----------------------------------------------------------------------------- homogeneous matrices (host)
def _hom(rows, n):
    """[n, k, k] matrix from a k x k table of per-sample vectors / python scalars."""
    k = len(rows)
    out = torch.zeros(n, k, k)
    for r in range(k):
        for c in range(k):
            out[:, r, c] = rows[r][c]
    return out


def _translate2(tx, ty):
    return _hom(((1, 0, tx), (0, 1, ty), (0, 0, 1)), tx.shape[0])


def _scale2(sx, sy):
    return _hom(((sx, 0, 0), (0, sy, 0), (0, 0, 1)), sx.shape[0])


def _rotate2(theta):
    c, s = torch.cos(theta), torch.sin(theta)
    return _hom(((c, -s, 0), (s, c, 0), (0, 0, 1)), theta.shape[0])


def _maybe(p, t, acc):
    """acc <- (t with probability p, identity otherwise) @ acc, per sample (non_leaking.py:143-148)."""
    n = t.shape[0]
    pick = torch.empty(n).bernoulli_(p).view(n, 1, 1)
    eye = torch.eye(t.shape[1]).expand_as(t)
    return (pick * t + (1 - pick) * eye) @ acc


def _choice(n, values):
    return torch.tensor(values)[torch.randint(high=len(values), size=(n,))]


def sample_affine(p, size, height, width):
    """Random geometric transform G[size, 3, 3] in normalised coordinates (non_leaking.py:151-207): x-flip,
    multiple-of-90 rotation, integer translation, isotropic scale, rotation, anisotropic scale, rotation,
    fractional translation — each applied with probability p (rotations: 1 - sqrt(1 - p) each)."""
    g = torch.eye(3).repeat(size, 1, 1)
    g = _maybe(p, _scale2(1 - 2.0 * _choice(size, (0, 1)), torch.ones(size)), g)
    g = _maybe(p, _rotate2(-math.pi / 2 * _choice(size, (0, 3))), g)
    u = torch.empty(size).uniform_(-0.125, 0.125)
    g = _maybe(p, _translate2(torch.round(u * width) / width, torch.round(u * height) / height), g)
    s = torch.empty(size).log_normal_(mean=0, std=0.2 * math.log(2))
    g = _maybe(p, _scale2(s, s), g)
    p_rot = 1 - math.sqrt(1 - p)
    g = _maybe(p_rot, _rotate2(-torch.empty(size).uniform_(-math.pi, math.pi)), g)
    s = torch.empty(size).log_normal_(mean=0, std=0.2 * math.log(2))
    g = _maybe(p, _scale2(s, 1 / s), g)
    g = _maybe(p_rot, _rotate2(-torch.empty(size).uniform_(-math.pi, math.pi)), g)
    t = torch.empty(size).normal_(0, 0.125)
    return _maybe(p, _translate2(t, t), g)


def sample_color(p, size):
    """Random colour transform C[size, 4, 4] on homogeneous RGB (non_leaking.py:210-241): brightness, contrast,
    luma flip, hue rotation about the grey axis, saturation — each with probability p."""
    c = torch.eye(4).repeat(size, 1, 1)
    a = 1 / math.sqrt(3)
    grey = torch.tensor((a, a, a, 0.0))
    proj = torch.outer(grey, grey)                         # projector on the luma axis (homogeneous 4x4)
    b = torch.empty(size).normal_(0, 0.2)
    c = _maybe(p, _hom(((1, 0, 0, b), (0, 1, 0, b), (0, 0, 1, b), (0, 0, 0, 1)), size), c)
    s = torch.empty(size).log_normal_(mean=0, std=0.5 * math.log(2))
    c = _maybe(p, _hom(((s, 0, 0, 0), (0, s, 0, 0), (0, 0, s, 0), (0, 0, 0, 1)), size), c)
    flip = _choice(size, (0, 1)).view(-1, 1, 1)
    c = _maybe(p, torch.eye(4) - 2 * proj * flip, c)
    theta = torch.empty(size).uniform_(-math.pi, math.pi)
    cos_t, sin_t = torch.cos(theta).view(-1, 1, 1), torch.sin(theta).view(-1, 1, 1)
    cross = torch.tensor(((0, -a, a), (a, 0, -a), (-a, a, 0)))
    rot = torch.eye(4).repeat(size, 1, 1)
    rot[:, :3, :3] = cos_t * torch.eye(3) + sin_t * cross + (1 - cos_t) * proj[:3, :3]   # Rodrigues
    c = _maybe(p, rot, c)
    s = torch.empty(size).log_normal_(mean=0, std=1 * math.log(2)).view(-1, 1, 1)
    return _maybe(p, proj + (torch.eye(4) - proj) * s, c)


# --------------------------------------------------------------------------------------- image path (device)
def _padding(g_inv, height, width):
    """Reflect padding that keeps the warped unit square inside the image (non_leaking.py:259-285):
    (x_low, x_high, y_low, y_high) in pixels, maximum over the batch."""
    corners = torch.tensor(((-1.0, -1, 1), (-1, 1, 1), (1, -1, 1), (1, 1, 1))).t()
    ext = g_inv[:, :2, :] @ corners                        # [n, 2, 4]
    size = torch.tensor((width, height))
    low = ((ext.min(-1).values + 1) * size).clamp(max=0).abs().ceil().max(0).values.to(torch.int64).tolist()
    high = (ext.max(-1).values * size - size).clamp(min=0).ceil().max(0).values.to(torch.int64).tolist()
    return low[0], high[0], low[1], high[1]


def random_apply_affine(img, p, G=None, antialiasing_kernel=SYM6):
    """non_leaking.py:316-371.  Returns (warped image, G)."""
    n, _, h, w = img.shape
    taps = torch.as_tensor(antialiasing_kernel, dtype=torch.float32)
    k2 = torch.outer(taps, taps).to(img)
    k2_flip = torch.flip(k2, (0, 1)).contiguous()
    len_k = taps.numel()
    pad_k = (len_k + 1) // 2
    given = G is not None
    while True:
        g = G if given else sample_affine(p, n, h, w)
        px1, px2, py1, py2 = _padding(torch.inverse(g.cpu().float()), h, w)
        try:
            padded = F.pad(img, (px1 + pad_k, px2 + pad_k, py1 + pad_k, py2 + pad_k), mode='reflect')
            break
        except RuntimeError:
            if given:
                raise                                       # the reference would retry forever with a fixed G
    wp, hp = padded.shape[3] - len_k + 1, padded.shape[2] - len_k + 1
    up = upfirdn2d(padded, k2_flip, up=2)
    # sampling grid: output pixel centres -> source coordinates of the 2x-upsampled padded image
    xs = torch.linspace(-2 * px1 / w - 1, 2 * (wp - px1) / w - 1, up.shape[3], device=img.device)
    ys = torch.linspace(-2 * py1 / h - 1, 2 * (hp - py1) / h - 1, up.shape[2], device=img.device)
    base = torch.stack((xs.view(1, -1).expand(up.shape[2], -1), ys.view(-1, 1).expand(-1, up.shape[3]),
                        torch.ones(up.shape[2], up.shape[3], device=img.device)), -1).to(up)      # [H2, W2, 3]
    m = torch.inverse(g.cpu().float())[:, :2, :].to(up)                                            # [n, 2, 3]
    grid = torch.einsum('hwk,njk->nhwj', base, m)
    grid = grid * torch.tensor((w / wp, h / hp), device=img.device) + torch.tensor(
        ((w + 2 * px1) / wp - 1, (h + 2 * py1) / hp - 1), device=img.device)
    warped = F.grid_sample(up, grid, mode='bilinear', align_corners=False, padding_mode='zeros')
    down = upfirdn2d(warped, k2, down=2)
    ey = down.shape[2] if -py2 - 1 == 0 else -py2 - 1
    ex = down.shape[3] if -px2 - 1 == 0 else -px2 - 1
    return down[:, :, py1:ey, px1:ex], g


def apply_color(img, mat):
    """img[n, 3, h, w] <- mat[:, :3, :3] @ rgb + mat[:, :3, 3]  (non_leaking.py:374-382)."""
    m = mat.to(img)
    return torch.einsum('nij,njhw->nihw', m[:, :3, :3], img) + m[:, :3, 3].view(-1, 3, 1, 1)


def random_apply_color(img, p, C=None):
    if C is None:
        C = sample_color(p, img.shape[0])
    return apply_color(img, C), C


def augment(img, p, transform_matrix=(None, None)):
    """Same signature and return value as the reference: (augmented image, (G, C))."""
    img, G = random_apply_affine(img, p, transform_matrix[0])
    img, C = random_apply_color(img, p, transform_matrix[1])
    return img, (G, C)
