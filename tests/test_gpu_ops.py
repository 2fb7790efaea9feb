"""GPU parity tests of the HIP ops (through the C ABI) against the CPU oracle and the
reference-derived golden vectors.  Run with `-m gpu` on an MI355X."""
import contextlib
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from rick_amd.synth import synth_state_dict, synth_tensor
from tests.cases import UPFIRDN_CASES, upfirdn_kernel

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


# ---------------------------------------------------------------------------- upfirdn2d
@pytest.mark.parametrize('case', UPFIRDN_CASES, ids=[c[0] for c in UPFIRDN_CASES])
def test_upfirdn2d_planar_bitexact(case, golden):
    """Planar path vs the scalar C oracle: bit-exact (same index math, same fmaf order)."""
    from oracle import c_ref
    from rick_amd.op import upfirdn2d
    tag, up, down, p0, p1, n, c, h, w, ks = case
    g = golden('ops')
    k = upfirdn_kernel(ks, up)
    x = synth_tensor(f'upfirdn/{tag}/x', (n, c, h, w))
    xd = x.to(DEV).requires_grad_(True)
    y = upfirdn2d(xd, k.to(DEV), up=up, down=down, pad=(p0, p1))
    yc = c_ref.upfirdn2d_c(x.numpy(), k.numpy(), (up, up), (down, down), (p0, p1, p0, p1))
    assert np.array_equal(y.detach().cpu().numpy(), yc), 'forward not bit-exact vs C oracle'
    assert rel_err(y, g[f'{tag}/y']) < 2e-6
    gy = synth_tensor(f'upfirdn/{tag}/gy', y.shape).to(DEV).requires_grad_(True)
    (gx,) = torch.autograd.grad(y, xd, gy, create_graph=True)
    assert rel_err(gx, g[f'{tag}/gx']) < 2e-6
    ggx = synth_tensor(f'upfirdn/{tag}/ggx', x.shape).to(DEV)
    (ggy,) = torch.autograd.grad(gx, gy, ggx)
    assert rel_err(ggy, g[f'{tag}/ggy']) < 2e-6


@pytest.mark.parametrize('cfg', [(1, 1, 1, 1, 2, 64, 17, 17), (1, 1, 2, 2, 2, 128, 16, 16), (1, 1, 1, 1, 1, 8, 9, 12),
                                 (1, 2, 1, 1, 2, 64, 16, 16), (2, 1, 2, 1, 1, 16, 8, 8), (1, 1, 1, 1, 3, 512, 9, 9),
                                 (1, 1, 2, 2, 1, 128, 256, 256)])
def test_upfirdn2d_nhwc_bitexact(cfg):
    from oracle import c_ref
    from rick_amd.op import upfirdn2d
    up, down, p0, p1, n, c, h, w = cfg
    k = upfirdn_kernel(4, up)
    x = synth_tensor(f'ufd_nhwc/{cfg}', (n, c, h, w))
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = upfirdn2d(xd, k.to(DEV), up=up, down=down, pad=(p0, p1))
    assert y.is_contiguous(memory_format=torch.channels_last)
    yc = c_ref.upfirdn2d_c(x.numpy(), k.numpy(), (up, up), (down, down), (p0, p1, p0, p1))
    assert np.array_equal(y.detach().cpu().numpy(), yc)
    gy = synth_tensor(f'ufd_nhwc/gy/{cfg}', y.shape)
    (gx,) = torch.autograd.grad(y, xd, gy.to(DEV))
    from oracle.ops_ref import upfirdn2d_ref
    xr = x.double().requires_grad_(True)
    (gxr,) = torch.autograd.grad(upfirdn2d_ref(xr, k.double(), up, down, (p0, p1)), xr, gy.double())
    assert rel_err(gx, gxr) < 2e-6


def test_upfirdn2d_device_dispatch_agrees():
    """op/upfirdn2d.py:145-149: CPU tensors take the host route, device tensors the HIP kernels — same values (the host route
    itself is pinned against the reference's CPU results in tests/test_op_cpu_dispatch.py); mixing devices is an error."""
    from rick_amd.op import upfirdn2d
    x = torch.randn(2, 3, 9, 9)
    k = torch.ones(4, 4) / 16
    y_host = upfirdn2d(x, k, up=2, pad=(2, 1))
    y_dev = upfirdn2d(x.cuda(), k.cuda(), up=2, pad=(2, 1))
    assert y_host.device.type == 'cpu' and y_dev.is_cuda
    assert rel_err(y_dev, y_host) < 2e-6
    with pytest.raises(RuntimeError):
        upfirdn2d(x.cuda(), k)


# ------------------------------------------------------------------------- fused act
@pytest.mark.parametrize('tag,shape', [('act2d', (3, 8)), ('act4d', (2, 5, 6, 6))])
def test_fused_leaky_relu_golden(tag, shape, golden):
    from rick_amd.op import fused_leaky_relu
    g = golden('ops')
    x = synth_tensor(f'act/{tag}/x', shape).to(DEV).requires_grad_(True)
    b = synth_tensor(f'act/{tag}/b', (shape[1],)).to(DEV).requires_grad_(True)
    y = fused_leaky_relu(x, b)
    assert rel_err(y, g[f'{tag}/y']) < 1e-6
    gy = synth_tensor(f'act/{tag}/gy', shape).to(DEV).requires_grad_(True)
    gx, gb = torch.autograd.grad(y, (x, b), gy, create_graph=True)
    assert rel_err(gx, g[f'{tag}/gx']) < 1e-6
    assert rel_err(gb, g[f'{tag}/gb']) < 1e-5
    ggx = synth_tensor(f'act/{tag}/ggx', shape).to(DEV)
    ggb = synth_tensor(f'act/{tag}/ggb', (shape[1],)).to(DEV)
    (ggy,) = torch.autograd.grad((gx, gb), gy, (ggx, ggb))
    assert rel_err(ggy, g[f'{tag}/ggy']) < 1e-6


@pytest.mark.parametrize('shape,nb', [((2, 64, 8, 8), 1), ((3, 128, 16, 16), 3), ((2, 512, 4, 4), 1), ((4, 12, 5, 5), 4)])
def test_noise_bias_act_vs_oracle(shape, nb):
    from oracle.ops_ref import fused_leaky_relu_ref
    from rick_amd.op import fused_noise_bias_act
    x = synth_tensor(f'nba/x/{shape}', shape)
    b = synth_tensor(f'nba/b/{shape}', (shape[1],))
    nz = synth_tensor(f'nba/n/{shape}', (nb, 1, shape[2], shape[3]))
    nw = synth_tensor(f'nba/w/{shape}', (1,))
    gy = synth_tensor(f'nba/gy/{shape}', shape)
    ref_in = [t.double().requires_grad_(True) for t in (x, b, nw)]
    yr = fused_leaky_relu_ref(ref_in[0] + ref_in[2] * nz.double(), ref_in[1])
    gr = torch.autograd.grad(yr, ref_in, gy.double(), create_graph=True)
    pl = sum((t ** 2).sum() for t in gr)
    ggr = torch.autograd.grad(pl, ref_in[0])
    dev_in = [t.to(DEV).requires_grad_(True) for t in (x, b, nw)]
    y = fused_noise_bias_act(dev_in[0], dev_in[1], nz.to(DEV), dev_in[2])
    assert rel_err(y, yr) < 1e-6
    gd = torch.autograd.grad(y, dev_in, gy.to(DEV), create_graph=True)
    for a, r in zip(gd, gr):
        assert rel_err(a, r) < 2e-5
    pld = sum((t ** 2).sum() for t in gd)
    (ggd,) = torch.autograd.grad(pld, dev_in[0])
    assert rel_err(ggd, ggr[0]) < 2e-5


# ------------------------------------------------------------------------------ conv
CONV_CASES = [
    # tag, N, Ci, Co, H, W, k, stride, pad
    ('s1_small', 2, 32, 128, 8, 8, 3, 1, 1),
    ('s1_odd_ch', 3, 16, 24, 8, 8, 3, 1, 1),
    ('s1_16', 2, 64, 64, 16, 16, 3, 1, 1),
    ('s1_rect', 1, 32, 32, 12, 20, 3, 1, 1),
    ('s1_4x4', 4, 64, 128, 4, 4, 3, 1, 1),
    ('s1_513', 2, 513, 128, 4, 4, 3, 1, 1),
    ('s1_big', 1, 128, 128, 64, 64, 3, 1, 1),
    ('s2_blur', 2, 32, 64, 17, 17, 3, 2, 0),
    ('s2_33', 1, 64, 128, 33, 33, 3, 2, 0),
    ('k1_s2', 2, 64, 32, 15, 15, 1, 2, 0),
    ('k1_s1', 2, 32, 64, 8, 8, 1, 1, 0),
    ('co256', 1, 32, 256, 8, 8, 3, 1, 1),
    ('co192', 2, 64, 192, 16, 16, 3, 1, 1),        # Co % 128 != 0: weight gradient must not take the whole-tile (FAST) form
    ('h18', 2, 32, 128, 18, 16, 3, 1, 1),          # 18 rows: not a multiple of the 16 x 4 position tile (two images: a whole-tile read of image 0 would run into image 1)
    ('ci48', 2, 48, 128, 16, 16, 3, 1, 1),         # Ci % 32 != 0
    ('w512', 2, 264, 500, 8, 8, 3, 1, 1),          # >= 32 (co tile, chunk) tiles: the weight gradient's second stage takes its tile form; ragged Co, Ci
    ('w512_k1', 2, 512, 256, 8, 8, 1, 1, 0),       # ... with one slice per (co, ci)
]


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d_vs_fp64(case):
    """conv, its data/weight gradients and one second-order term vs torch CPU fp64.
    fp16 hi/lo split: fp32-grade (measured 2-4e-7, the level of an fp32 CPU convolution), tolerance 2e-6 relative to the
    output max."""
    from rick_amd import op
    tag, N, Ci, Co, H, W, k, s, p = case
    x = synth_tensor(f'conv/{tag}/x', (N, Ci, H, W))
    w = synth_tensor(f'conv/{tag}/w', (Co, Ci, k, k))
    wscale = 1 / math.sqrt(Ci * k * k)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr * wscale, stride=s, padding=p)
    gy = synth_tensor(f'conv/{tag}/gy', yr.shape)
    gxr, gwr = torch.autograd.grad(yr, (xr, wr), gy.double(), create_graph=True)
    plr = gxr.pow(2).sum() + gwr.pow(2).sum()
    ggr = torch.autograd.grad(plr, (xr, wr))
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = op.conv2d(xd, wd, s, p, wscale=wscale)
    assert y.shape == yr.shape
    assert rel_err(y, yr) < 2e-6, 'fprop'
    gx, gw = torch.autograd.grad(y, (xd, wd), gy.to(DEV), create_graph=True)
    assert rel_err(gx, gxr) < 2e-6, 'dgrad'
    assert rel_err(gw, gwr) < 2e-6, 'wgrad'
    pl = gx.pow(2).sum() + gw.pow(2).sum()
    gg = torch.autograd.grad(pl, (xd, wd))
    assert rel_err(gg[0], ggr[0]) < 5e-6, 'second-order d/dx'
    assert rel_err(gg[1], ggr[1]) < 5e-6, 'second-order d/dw'


@contextlib.contextmanager
def _conv_tuning(**kv):
    """rick_conv_tuning (include/rick_hip.h) for the duration of a test: which kernel FORM a launch takes, never its values."""
    from rick_amd._lib import lib
    keys = {'igemm_w8': 0, 'igemm_w8_minblk': 1, 'splitk_fused': 2, 'ufd_tile16': 3, 'igemm_s2w8': 4}
    prev = {k: lib.rick_conv_tuning(keys[k], v) for k, v in kv.items()}
    try:
        yield
    finally:
        for k, v in prev.items():
            lib.rick_conv_tuning(keys[k], v)


@pytest.mark.parametrize('cfg', [(1, 1, 2, 64, 64, 64), (2, 2, 1, 128, 65, 40), (1, 1, 3, 192, 33, 129)])
def test_upfirdn2d_nhwc_tile16_equals_tile8_and_c_oracle(cfg):
    """The 4x4 FIR's 16 x 16 x 32-channel tile (round 6; model_probe_tune.py:609-629, op/upfirdn2d_kernel.cu:107-207 are what
    it replaces) against the 8 x 8 x 64-channel tile and the scalar C oracle: every output accumulates its 16 taps y-outer /
    x-inner either way — bit-equal, ragged edge tiles and 3 x 64 channels included; backward (the adjoint FIR) too."""
    from oracle import c_ref
    from rick_amd.op import upfirdn2d
    p0, p1, n, c, h, w = cfg
    k = upfirdn_kernel(4, 1)
    x = synth_tensor(f'ufd16/{cfg}', (n, c, h, w))
    outs = {}
    for mode in (0, 1):
        with _conv_tuning(ufd_tile16=mode):
            xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            y = upfirdn2d(xd, k.to(DEV), up=1, down=1, pad=(p0, p1))
            gy = synth_tensor(f'ufd16/{cfg}/gy', y.shape).to(DEV).contiguous(memory_format=torch.channels_last)
            (gx,) = torch.autograd.grad(y, xd, gy)
            torch.cuda.synchronize()
            outs[mode] = (y.detach(), gx)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    yc = c_ref.upfirdn2d_c(x.numpy(), k.numpy(), (1, 1), (1, 1), (p0, p1, p0, p1))
    assert np.array_equal(outs[1][0].cpu().numpy(), yc)


@pytest.mark.parametrize('case', [('ragged_co', 2, 128, 192, 32, 32), ('ragged_tiles', 1, 160, 128, 40, 24),
                                  ('three_images', 3, 256, 128, 16, 48)], ids=lambda c: c[0])
def test_igemm_eight_wave_form_vs_fp64(case):
    """The eight-wave 128 co x 256 position igemm block (conv.hip, igemm_body NW = 8: 4-slot LDS-DMA weight ring, two patch
    buffers, fragments read across the barrier), forced onto small geometries: forward with modulation scales, plain forward
    and the stride-1 data gradient (model_probe_tune.py:122,278-282) against torch CPU fp64 at the four-wave form's
    tolerance — ragged co tiles, position tiles hanging over the image, channel counts that are no multiple of 32."""
    from rick_amd.op import conv as cv
    tag, n, ci, co, h, w = case
    x = synth_tensor(f'w8/{tag}/x', (n, ci, h, w)) * torch.exp2(torch.randint(-6, 3, (n, ci, 1, 1), generator=torch.Generator().manual_seed(1)).float())
    wt = synth_tensor(f'w8/{tag}/w', (co, ci, 3, 3))
    si, so = synth_tensor(f'w8/{tag}/si', (n, ci)).abs() + 0.5, synth_tensor(f'w8/{tag}/so', (n, co)).abs() + 0.5
    gy = synth_tensor(f'w8/{tag}/gy', (n, co, h, w)) * 1e-4
    ref = F.conv2d(x.double() * si.double()[:, :, None, None], wt.double(), padding=1) * so.double()[:, :, None, None]
    ref0 = F.conv2d(x.double(), wt.double(), padding=1)
    refT = F.conv_transpose2d(gy.double(), wt.double(), padding=1)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    gd = gy.to(DEV).contiguous(memory_format=torch.channels_last)
    wp, wpT = cv._pack(wt.to(DEV), 1.0), cv._pack(wt.to(DEV).transpose(0, 1), 1.0)
    with _conv_tuning(igemm_w8=2, igemm_w8_minblk=1):
        y = cv._conv_launch(xd, wp, co, 3, 3, 1, 1, iscale=si.to(DEV), oscale=so.to(DEV))
        y0 = cv._conv_launch(xd, wp, co, 3, 3, 1, 1)
        gx = cv._convT_launch(gd, wpT, ci, 3, 3, 1, 1, (h, w))
        torch.cuda.synchronize()
    assert rel_err(y, ref) < 2e-6, 'modulated forward'
    assert rel_err(y0, ref0) < 2e-6, 'plain forward'
    assert rel_err(gx, refT) < 2e-6, 'data gradient'


@pytest.mark.parametrize('case', [('co256', 2, 128, 256, 33), ('ragged_tiles', 1, 160, 256, 41), ('three_images', 3, 256, 512, 17)],
                         ids=lambda c: c[0])
def test_igemm_stride2_eight_wave_form_vs_fp64(case):
    """The eight-wave STRIDE-2 igemm block (conv.hip, igemm_body WDMA = 5: 256 co x 128 positions, two weight tiles per k-step
    by LDS-DMA, the 33 x 17 patch staged once for 256 channels) — the discriminator's downsampling 3x3 convolutions
    (model_probe_tune.py:609-629) — forced onto small geometries, with demodulation-style output scales, against torch CPU fp64 at
    the four-wave form's tolerance: position tiles hanging over the image, channel counts that are no multiple of 32."""
    from rick_amd.op import conv as cv
    tag, n, ci, co, r = case
    x = synth_tensor(f's2w8/{tag}/x', (n, ci, r, r)) * torch.exp2(torch.randint(-6, 3, (n, ci, 1, 1), generator=torch.Generator().manual_seed(2)).float())
    wt = synth_tensor(f's2w8/{tag}/w', (co, ci, 3, 3))
    si, so = synth_tensor(f's2w8/{tag}/si', (n, ci)).abs() + 0.5, synth_tensor(f's2w8/{tag}/so', (n, co)).abs() + 0.5
    ref = F.conv2d(x.double() * si.double()[:, :, None, None], wt.double(), stride=2) * so.double()[:, :, None, None]
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    wp = cv._pack(wt.to(DEV), 1.0)
    with _conv_tuning(igemm_s2w8=2, igemm_w8_minblk=1):
        y = cv._conv_launch(xd, wp, co, 3, 3, 2, 0, iscale=si.to(DEV), oscale=so.to(DEV))
        y0 = cv._conv_launch(xd, wp, co, 3, 3, 2, 0)
        torch.cuda.synchronize()
    assert rel_err(y, ref) < 2e-6, 'scaled forward'
    assert rel_err(y0, F.conv2d(x.double(), wt.double(), stride=2)) < 2e-6, 'plain forward'


def test_igemm_stride2_eight_wave_form_bit_equal_on_split_images():
    """Same MFMA order per accumulator as the four-wave 128 x 64 blocks: bit-equal on split images (geometries that fill the chip
    with four-wave blocks, so neither form splits K), the fused bias + LeakyReLU tail included; fp32 operands differ by the
    per-block exponents only.  (The 256-px model tests — goldens, loop body, teacher-forced steps — run the discriminator's
    128 -> 256 and 256 -> 512 downsampling convolutions through this form.)"""
    from rick_amd.op import conv as cv, split as sp
    n, ci, co, r = 8, 128, 256, 129
    x = torch.randn(n, ci, r, r, device=DEV, generator=torch.Generator(DEV).manual_seed(7)).contiguous(memory_format=torch.channels_last)
    wt = synth_tensor('s2w8/eq/w', (co, ci, 3, 3)).to(DEV)
    bias = synth_tensor('s2w8/eq/b', (co,)).to(DEV)
    wp = cv._pack(wt, 1.0)
    xs = sp.split_pack(x)
    epi = cv._epilogue(bias, None, None, 0.2, 2 ** 0.5)
    outs = {}
    for mode in (0, 2):
        with _conv_tuning(igemm_s2w8=mode):
            outs[mode] = (cv._conv_launch(None, wp, co, 3, 3, 2, 0, epi=epi, x_split=xs), cv._conv_launch(x, wp, co, 3, 3, 2, 0))
            torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[2][0]), 'forward on a split image'
    assert rel_err(outs[2][1], outs[0][1]) < 2e-6, 'fp32 operands: per-block exponents only'


def test_igemm_eight_wave_form_bit_equal_on_split_images():
    """Same MFMA order per accumulator as the four-wave blocks: on split images (one exponent per tensor, no per-block
    sampling) the two forms agree bit for bit — forward and stride-1 data gradient on a geometry that fills the chip with
    four-wave blocks (no split-K there, so both forms sum the channel chunks in one chain); the fused bias + noise +
    LeakyReLU tail included.  On fp32 operands they differ by the per-block operand exponents only."""
    from rick_amd.op import conv as cv, split as sp
    n, ci, co, r = 4, 128, 128, 128
    x = (torch.randn(n, ci, r, r, device=DEV, generator=torch.Generator(DEV).manual_seed(5))).contiguous(memory_format=torch.channels_last)
    gy = (torch.randn(n, co, r, r, device=DEV, generator=torch.Generator(DEV).manual_seed(6)) * 1e-3).contiguous(memory_format=torch.channels_last)
    wt = synth_tensor('w8/eq/w', (co, ci, 3, 3)).to(DEV)
    bias, so = synth_tensor('w8/eq/b', (co,)).to(DEV), synth_tensor('w8/eq/so', (n, co)).abs().to(DEV) + 0.5
    noise, nw = synth_tensor('w8/eq/noise', (1, 1, r, r)).to(DEV), torch.full((1,), 0.3, device=DEV)
    wp, wpT = cv._pack(wt, 1.0), cv._pack(wt.transpose(0, 1), 1.0)
    xs, gs = sp.split_pack(x), sp.split_pack(gy)
    epi = cv._epilogue(bias, noise, nw, 0.2, 2 ** 0.5)
    outs = {}
    for mode in (0, 2):
        with _conv_tuning(igemm_w8=mode):
            outs[mode] = (cv._conv_launch(None, wp, co, 3, 3, 1, 1, oscale=so, epi=epi, x_split=xs),
                          cv._convT_launch(None, wpT, ci, 3, 3, 1, 1, (r, r), x_split=gs),
                          cv._conv_launch(x, wp, co, 3, 3, 1, 1))
            torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[2][0]), 'forward on a split image'
    assert torch.equal(outs[0][1], outs[2][1]), 'data gradient on a split image'
    assert rel_err(outs[2][2], outs[0][2]) < 2e-6, 'fp32 operands: per-block exponents only'
    assert not torch.equal(outs[0][2], torch.zeros_like(outs[0][2]))


@pytest.mark.parametrize('case', [('s1_4', 4, 512, 512, 4, 3, 1, 1), ('s1_8', 4, 512, 512, 8, 3, 1, 1), ('s1_16', 8, 512, 512, 16, 3, 1, 1),
                                  ('s2_17', 4, 512, 512, 17, 3, 2, 0), ('k1_8', 8, 512, 512, 8, 1, 1, 0), ('ragged', 3, 160, 200, 9, 3, 1, 1)],
                         ids=lambda c: c[0])
def test_splitk_fixup_inside_the_launch_equals_second_stage_kernel(case):
    """Split-K launches of the igemm family (the 4^2 ... 16^2 layers, model_probe_tune.py:400-410): the block that arrives last
    for an output tile sums the partial tiles in split order and applies alpha / demodulation / the fused tail — bit-equal to
    the stand-alone second-stage kernel (rick_conv_tuning RICK_TUNE_SPLITK_FUSED = 0), three launches in a row on the same
    self-resetting tickets; and both within 2e-6 of fp64."""
    from rick_amd.op import conv as cv
    tag, n, ci, co, r, k, s, p = case
    x = synth_tensor(f'skf/{tag}/x', (n, ci, r, r)).to(DEV).contiguous(memory_format=torch.channels_last)
    wt = synth_tensor(f'skf/{tag}/w', (co, ci, k, k)).to(DEV)
    si, so = synth_tensor(f'skf/{tag}/si', (n, ci)).abs().to(DEV) + 0.5, synth_tensor(f'skf/{tag}/so', (n, co)).abs().to(DEV) + 0.5
    bias = synth_tensor(f'skf/{tag}/b', (co,)).to(DEV)
    ro = (r + 2 * p - k) // s + 1
    noise, nw = synth_tensor(f'skf/{tag}/noise', (1, 1, ro, ro)).to(DEV), torch.full((1,), 0.3, device=DEV)
    wp = cv._pack(wt, 0.05)
    epi = cv._epilogue(bias, noise, nw, 0.2, 2 ** 0.5) if co % 4 == 0 else None
    outs = {}
    for fused in (0, 1):
        with _conv_tuning(splitk_fused=fused):
            outs[fused] = [(cv._conv_launch(x, wp, co, k, k, s, p, iscale=si, oscale=so, alpha=0.05, epi=epi), cv._conv_launch(x, wp, co, k, k, s, p))
                           for _ in range(3)]
            torch.cuda.synchronize()
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(outs[1][0][0], outs[1][2][0])
    ref = F.conv2d(x.double().cpu(), wt.double().cpu() * 0.05, stride=s, padding=p)
    assert rel_err(outs[1][0][1], ref) < 2e-6


@pytest.mark.parametrize('ci,co', [(512, 256), (64, 96)], ids=['tile_stage', 'element_stage'])
def test_wgrad_accumulates_into_both_parameter_layouts(ci, co):
    """rick_conv_wgrad_f32 with accumulate = 1 (the trainer's gradient sink): [O, I, kh, kw] and the transposed-conv
    parameter layout [I, O, kh, kw] receive exactly the stand-alone result, on both second-stage kernels (>= 32 tiles:
    the LDS-transposing one)."""
    from rick_amd.op import conv as cv
    a_cpu, b_cpu = synth_tensor(f'wacc/{ci}/a', (2, co, 8, 8)), synth_tensor(f'wacc/{ci}/b', (2, ci, 8, 8))
    a = a_cpu.to(DEV).contiguous(memory_format=torch.channels_last)        # output-side operand
    b = b_cpu.to(DEV).contiguous(memory_format=torch.channels_last)        # input-side operand
    gw = cv._wgrad_launch(a, b, 3, 3, 1, 1, alpha=0.5)
    assert gw.shape == (co, ci, 3, 3)
    w0 = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    ref = torch.autograd.grad(F.conv2d(b_cpu.double(), w0, padding=1), w0, a_cpu.double())[0] * 0.5
    assert rel_err(gw, ref) < 2e-6
    base = synth_tensor(f'wacc/{ci}/base', (co, ci, 3, 3)).to(DEV)
    sink = base.clone()
    assert cv._wgrad_launch(a, b, 3, 3, 1, 1, alpha=0.5, out=sink) is None
    assert torch.equal(sink, base + gw)
    sink_t = base.transpose(0, 1).contiguous()
    cv._wgrad_launch(a, b, 3, 3, 1, 1, alpha=0.5, out=sink_t, transposed=True)
    assert torch.equal(sink_t, (base + gw).transpose(0, 1))


@pytest.mark.parametrize('case', [('t_4', 2, 32, 64, 4, 4), ('t_8', 3, 64, 32, 8, 8), ('t_16', 1, 128, 128, 16, 16),
                                  ('t_5', 2, 16, 24, 5, 7)], ids=lambda c: c[0])
def test_conv_transpose2d_vs_fp64(case):
    from rick_amd import op
    tag, N, Ci, Co, H, W = case
    x = synth_tensor(f'convT/{tag}/x', (N, Ci, H, W))
    w = synth_tensor(f'convT/{tag}/w', (Co, Ci, 3, 3))
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr.transpose(0, 1) * 0.1, stride=2, padding=0)
    gy = synth_tensor(f'convT/{tag}/gy', yr.shape)
    gxr, gwr = torch.autograd.grad(yr, (xr, wr), gy.double(), create_graph=True)
    ggr = torch.autograd.grad(gxr.pow(2).sum() + gwr.pow(2).sum(), (xr, wr))
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = op.conv_transpose2d(xd, wd, 2, 0, wscale=0.1)
    assert y.shape == yr.shape
    assert rel_err(y, yr) < 2e-6
    gx, gw = torch.autograd.grad(y, (xd, wd), gy.to(DEV), create_graph=True)
    assert rel_err(gx, gxr) < 2e-6
    assert rel_err(gw, gwr) < 2e-6
    gg = torch.autograd.grad(gx.pow(2).sum() + gw.pow(2).sum(), (xd, wd))
    assert rel_err(gg[0], ggr[0]) < 5e-6
    assert rel_err(gg[1], ggr[1]) < 5e-6


CT2_CASES = [  # tag, N, Ci, Co, IH, IW, crop (output 2*IH instead of 2*IH+1), scales
    ('g4', 4, 512, 512, 4, 4, False, True),          # G convs.0: split-K over all 16 chunks, 4 images per tile
    ('g8', 4, 512, 512, 8, 8, False, True),
    ('g16_n8', 8, 512, 512, 16, 16, False, False),    # D dgrad shape at N = 8 (cat(fake, real))
    ('g64', 4, 512, 256, 64, 64, False, True),        # G convs.8 (512 -> 256 @64^2 -> 129^2): the benchmarked layer
    ('g128_n8', 8, 256, 128, 128, 128, False, False),  # D conv2 dgrad 128^2 -> 257^2, N = 8
    ('odd', 3, 36, 20, 5, 7, False, True),            # ragged channels (Ci, Co % 4 == 0 but not % 32 / % 128), odd sizes
    ('crop', 2, 64, 32, 9, 6, True, False),           # even-sized target: last output row / column cropped
]


CT2_SUBBLOCKS = {'g64': (True, 4), 'g128_n8': (True, 2), 'g8': (False, 2), 'g4': (False, 4)}   # tag -> (whole-tile launch too, blocks per tile)


@pytest.mark.parametrize('case', CT2_CASES, ids=[c[0] for c in CT2_CASES])
def test_convt2_single_staging_kernel(case):
    """rick_convt2_f32 (all four parity classes from one staged patch) vs F.conv_transpose2d in fp64 and vs the generic
    multi-class launch it replaces, at the real layer shapes incl. modulation / demodulation scales."""
    from rick_amd.op import conv as cv
    tag, N, Ci, Co, IH, IW, crop, scales = case
    x = synth_tensor(f'ct2/{tag}/x', (N, Ci, IH, IW))
    w = synth_tensor(f'ct2/{tag}/w', (Co, Ci, 3, 3))
    si = (synth_tensor(f'ct2/{tag}/si', (N, Ci)) * 0.5 + 1.0) if scales else None
    so = (synth_tensor(f'ct2/{tag}/so', (N, Co)) * 0.5 + 1.0) if scales else None
    wscale = 1.0 / (Ci * 9) ** 0.5
    xs = x.double() * (si.double()[:, :, None, None] if scales else 1.0)
    ref = F.conv_transpose2d(xs, w.double().transpose(0, 1) * wscale, stride=2)
    if scales:
        ref = ref * so.double()[:, :, None, None]
    OH, OW = (2 * IH, 2 * IW) if crop else (2 * IH + 1, 2 * IW + 1)
    ref = ref[:, :, :OH, :OW]
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    wp = cv._pack(w.to(DEV), wscale)
    sid, sod = (si.to(DEV), so.to(DEV)) if scales else (None, None)
    assert cv._USE_CT2
    if tag in CT2_SUBBLOCKS:     # these grids end in a partly filled round: its tiles run as 2 / 4 blocks of 4 / 2 fragment columns
        import ctypes
        from rick_amd._lib import lib
        plan = (ctypes.c_int * 8)()
        assert lib.rick_convt2_plan(N, IH, IW, Ci, Co, OH, OW, plan) == 0
        items, nfull, subq = plan[3] * plan[4], plan[6], plan[7]
        assert (nfull > 0, subq) == CT2_SUBBLOCKS[tag] and nfull < items, list(plan)
    y = cv._convT_launch(xd, wp, Co, 3, 3, 2, 0, (OH, OW), iscale=sid, oscale=sod)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < 2e-6
    y2 = cv._convT_launch(xd, wp, Co, 3, 3, 2, 0, (OH, OW), iscale=sid, oscale=sod)
    assert torch.equal(y, y2), 'not deterministic'
    cv._USE_CT2 = False
    try:
        y_old = cv._convT_launch(xd, wp, Co, 3, 3, 2, 0, (OH, OW), iscale=sid, oscale=sod)
    finally:
        cv._USE_CT2 = True
    assert rel_err(y, y_old.double()) < 2e-6      # same products, different summation order over channel chunks


@pytest.mark.parametrize('case', ['small_first_chunk', 'spike'])
def test_conv_operand_exponent_covers_all_chunks(case):
    """Blocks that walk several channel chunks (no split-K: 512 blocks of 4 chunks here) take the operand exponent from
    the first chunk AND from samples of the others: first 32 channels 1e-6 of the rest (pruned / dead channels) gave inf
    with a first-chunk exponent.  A lone value 1e7 x everything else that no sample happens to see saturates at the fp16
    maximum instead of turning into inf / NaN (MODE.FP16_OVFL): finite everywhere, exact in the blocks without it."""
    from rick_amd import op
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 128, 64, 64, generator=g)
    w = torch.randn(512, 128, 3, 3, generator=g)
    gy = torch.randn(4, 512, 64, 64, generator=g)
    wscale = 1 / math.sqrt(128 * 9)
    if case == 'small_first_chunk':
        x[:, :32] *= 1e-6
        gy[:, :32] *= 1e-6
    else:
        x[1, 77, 40, 21] = 1e7
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = op.conv2d(xd, wd, 1, 1, wscale=wscale)
    gx, gw = torch.autograd.grad(y, (xd, wd), gy.to(DEV))
    yt = op.conv_transpose2d(xd[:, :, :32, :32].contiguous(), wd[:128], 2, 0, wscale=wscale)
    for name, t in (('y', y), ('gx', gx), ('gw', gw), ('yt', yt)):
        assert torch.isfinite(t).all(), name
    xr = x.double().requires_grad_(True)
    yr = F.conv2d(xr, w.double() * wscale, padding=1)
    (gxr,) = torch.autograd.grad(yr, xr, gy.double())
    if case == 'small_first_chunk':
        assert rel_err(y, yr) < 2e-6, rel_err(y, yr)
        assert rel_err(gx, gxr) < 2e-6, rel_err(gx, gxr)
        ytr = F.conv_transpose2d(x[:, :, :32, :32].double(), (w[:128].double() * wscale).transpose(0, 1), stride=2)
        assert rel_err(yt, ytr) < 2e-6, rel_err(yt, ytr)
    else:
        # blocks that hold the spike either saw it in their sample (exponent from 1e7: their O(1) values keep 2^-27 of the
        # block maximum, the documented block-exponent behaviour) or saturate it; every other image is untouched
        err = (y.cpu().double() - yr).abs()
        others = [0, 2, 3]
        assert float(err[others].max()) < 2e-6 * float(yr[others].abs().max())
        reach = torch.zeros(64, 64, dtype=torch.bool)
        reach[39:42, 20:23] = True                       # outputs the spike reaches
        assert float(err[1][:, ~reach].max()) < 2e-6 * float(yr.abs().max())
        assert rel_err(gx, gxr) < 2e-6                    # (the data gradient does not read x)


def test_conv_fp16_single_pass_is_coarser():
    """split=1 (plain fp16 MFMA) is a speed option, not the parity path: ~2^-12 per product instead of ~2^-22."""
    from rick_amd import op
    x = synth_tensor('convp/x', (2, 64, 16, 16))
    w = synth_tensor('convp/w', (64, 64, 3, 3))
    yr = F.conv2d(x.double(), w.double() / 24, padding=1)
    op.set_precision('fp16')
    try:
        y1 = op.conv2d(x.to(DEV), w.to(DEV), 1, 1, wscale=1 / 24)
    finally:
        op.set_precision('fp16x3')
    y3 = op.conv2d(x.to(DEV), w.to(DEV), 1, 1, wscale=1 / 24)
    e1, e3 = rel_err(y1, yr), rel_err(y3, yr)
    assert e3 < 1e-6 < 2e-5 < e1 < 3e-3, (e1, e3)


# --------------------------------------------------------------------------- mod conv
@pytest.mark.parametrize('tag', ['plain', 'up', 'rgb'])
@pytest.mark.parametrize('mode', ['fused', 'second_order'])
def test_modulated_conv_golden(tag, mode, golden):
    """ModulatedConv2d module vs the reference module's outputs (tests/golden/layers.npz)."""
    from rick_amd import op
    from rick_amd.models import ModulatedConv2d
    g = golden('layers')
    B, CI, CO, R, SD = 3, 16, 24, 8, 32
    co = 3 if tag == 'rgb' else CO
    kw = dict(kernel_size=1, demodulate=False) if tag == 'rgb' else dict(kernel_size=3, upsample=(tag == 'up'))
    m = ModulatedConv2d(CI, co, style_dim=SD, **kw)
    sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()})
    m.load_state_dict(sd, strict=False)
    m = m.to(DEV)
    x = synth_tensor(f'modconv/{tag}/x', (B, CI, R, R)).to(DEV).requires_grad_(True)
    s = synth_tensor(f'modconv/{tag}/s', (B, SD)).to(DEV).requires_grad_(True)
    gy = synth_tensor(f'modconv/{tag}/gy', g[f'{tag}/y'].shape).to(DEV)
    params = [m.weight, m.modulation.weight, m.modulation.bias]
    with op.second_order(mode == 'second_order'):
        y = m(x, s)
        assert rel_err(y, g[f'{tag}/y']) < 3e-5
        grads = torch.autograd.grad(y, [x, s] + params, gy, create_graph=(mode == 'second_order'))
        for n, gr in zip(('gx', 'gs', 'gw', 'gmw', 'gmb'), grads):
            assert rel_err(gr, g[f'{tag}/{n}']) < 5e-5, n
        if mode == 'second_order':
            pl = grads[1].pow(2).sum()
            assert rel_err(pl, g[f'{tag}/pl']) < 1e-4
            gg = torch.autograd.grad(pl, [x, m.weight, s], allow_unused=True)
            for n, gr, t in zip(('pl_gx', 'pl_gw', 'pl_gs'), gg, (x, m.weight, s)):
                gr = torch.zeros_like(t) if gr is None else gr
                ref = g[f'{tag}/{n}']
                if np.abs(ref).max() == 0:
                    assert float(gr.abs().max()) < 1e-6
                else:
                    assert rel_err(gr, ref) < 2e-4, n


# ------------------------------------------------------------------------ small ops
def test_thin_ops_vs_einsum():
    from rick_amd import op
    N, C, H, W, J = 3, 64, 9, 7, 3
    x = synth_tensor('thin/x', (N, C, H, W))
    Wm = synth_tensor('thin/W', (N, J, C))
    t = synth_tensor('thin/t', (N, J, H, W))
    xd, Wd, td = (a.to(DEV).requires_grad_(True) for a in (x, Wm, t))
    y = op.thin_fwd(xd, Wd)
    yr = torch.einsum('nchw,njc->njhw', x.double(), Wm.double())
    assert rel_err(y, yr) < 1e-5
    gx, gW = torch.autograd.grad(y, (xd, Wd), td, create_graph=True)
    assert rel_err(gx, torch.einsum('njhw,njc->nchw', t.double(), Wm.double())) < 1e-5
    assert rel_err(gW, torch.einsum('njhw,nchw->njc', t.double(), x.double())) < 1e-5
    (ggt,) = torch.autograd.grad(gx.pow(2).sum(), td)
    gxr = torch.einsum('njhw,njc->nchw', t.double(), Wm.double())
    assert rel_err(ggt, 2 * torch.einsum('nchw,njc->njhw', gxr, Wm.double())) < 2e-5
    # shared W (batch 1): discriminator input conv
    W1 = synth_tensor('thin/W1', (1, J, C)).to(DEV).requires_grad_(True)
    z = op.thin_bwdx(td, W1)
    assert rel_err(z, torch.einsum('njhw,jc->nchw', t.double(), W1[0].double().cpu())) < 1e-5
    (gW1,) = torch.autograd.grad(z, W1, xd.detach())
    assert rel_err(gW1[0], torch.einsum('njhw,nchw->jc', t.double(), x.double())) < 1e-5


def test_chan_scale_hw_dot_add_scale():
    from rick_amd import op
    x = synth_tensor('cs/x', (3, 64, 5, 7))
    s = synth_tensor('cs/s', (3, 64))
    xd, sd = x.to(DEV).requires_grad_(True), s.to(DEV).requires_grad_(True)
    y = op.chan_scale(xd, sd)
    assert rel_err(y, x * s[:, :, None, None]) < 1e-6
    g = synth_tensor('cs/g', x.shape).to(DEV)
    gx, gs = torch.autograd.grad(y, (xd, sd), g, create_graph=True)
    assert rel_err(gs, (g.cpu() * x).sum((2, 3))) < 1e-5
    (gg,) = torch.autograd.grad(gs.pow(2).sum(), xd)
    assert rel_err(gg, 2 * (g.cpu() * x).sum((2, 3))[:, :, None, None] * g.cpu()) < 1e-5
    a, b = synth_tensor('as/a', (2, 32, 6, 6)), synth_tensor('as/b', (2, 32, 6, 6))
    z = op.add_scale(a.to(DEV), b.to(DEV), 0.7071)
    assert rel_err(z, (a + b) * 0.7071) < 1e-6


@pytest.mark.parametrize('B', [1, 2, 4])
def test_minibatch_stddev(B):
    from oracle.ops_ref import minibatch_stddev_ref
    from rick_amd import op
    x = synth_tensor(f'mbstd/{B}', (B, 512, 4, 4))
    xr = x.double().requires_grad_(True)
    yr = minibatch_stddev_ref(xr)
    g = synth_tensor(f'mbstd/g/{B}', yr.shape)
    (gr,) = torch.autograd.grad(yr, xr, g.double())
    for so in (False, True):
        xd = x.to(DEV).requires_grad_(True)
        y = op.minibatch_stddev(xd, second_order=so)
        assert rel_err(y, yr) < 1e-5
        (gd,) = torch.autograd.grad(y, xd, g.to(DEV))
        assert rel_err(gd, gr) < 1e-4 if B > 1 else True
    # two concatenated calls == two separate calls
    x2 = synth_tensor(f'mbstd2/{B}', (2 * B, 512, 4, 4))
    g2 = synth_tensor(f'mbstd2/g/{B}', (2 * B, 513, 4, 4))
    for so in (False, True):
        xd = x2.to(DEV).requires_grad_(True)
        y = op.minibatch_stddev(xd, second_order=so, calls=2)
        (gd,) = torch.autograd.grad(y, xd, g2.to(DEV))
        xr = x2.double().requires_grad_(True)
        yr2 = torch.cat([minibatch_stddev_ref(c) for c in xr.chunk(2)], 0)
        (gr2,) = torch.autograd.grad(yr2, xr, g2.double())
        assert rel_err(y, yr2) < 1e-5
        if B > 1:
            assert rel_err(gd, gr2) < 1e-4


def test_fisher_and_optimizer_kernels():
    from oracle.train_ref import adam_step_ref
    from rick_amd import _lib
    lib, ptr, sp = _lib.lib, _lib.ptr, _lib.stream_ptr
    n = 100003
    g = synth_tensor('opt/g', (n,))
    acc = synth_tensor('opt/acc', (n,)).abs()
    accd, gd = acc.to(DEV), g.to(DEV)
    _lib.check(lib.rick_sq_accumulate_f32(ptr(accd), ptr(gd), n, sp()), 'sq')
    assert rel_err(accd, acc.double() + g.double() ** 2) < 1e-6
    # per-filter mean of a [1, Co, Ci, 3, 3] tensor over (0,2,3,4)  (train_dynamic_update_prune.py:282)
    f = synth_tensor('opt/f', (1, 24, 16, 3, 3)).abs()
    fd, out = f.to(DEV), torch.empty(24, device=DEV)
    _lib.check(lib.rick_filter_reduce_f32(ptr(fd), ptr(out), 1, 0, 24, 16 * 9, 16 * 9, 1.0 / (16 * 9), sp()), 'fr')
    assert rel_err(out, f.double().mean((0, 2, 3, 4))) < 1e-6
    # modulation weight [Ci, style] mean over dim 1
    f2 = synth_tensor('opt/f2', (16, 512)).abs()
    out2 = torch.empty(16, device=DEV)
    _lib.check(lib.rick_filter_reduce_f32(ptr(f2.to(DEV)), ptr(out2), 1, 0, 16, 512, 512, 1.0 / 512, sp()), 'fr2')
    assert rel_err(out2, f2.double().mean(1)) < 1e-6
    # masked Adam, 3 steps, beta1 = 0 as configured by the reference (train...:913-931)
    p = synth_tensor('opt/p', (n,))
    mask = (torch.arange(n) % 7 == 0).to(torch.uint8) + 2 * (torch.arange(n) % 11 == 0).to(torch.uint8)
    pd, md, vd, maskd = p.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), mask.to(DEV)
    pr, mr, vr = p.double(), torch.zeros(n).double(), torch.zeros(n).double()
    lr, b1, b2 = 0.002 * 0.8, 0.0, 0.99 ** 0.8
    for step in range(1, 4):
        gs = synth_tensor(f'opt/g{step}', (n,))
        gdd = gs.to(DEV)
        _lib.check(lib.rick_masked_adam_f32(ptr(pd), ptr(gdd), ptr(md), ptr(vd), ptr(maskd), n, lr, b1, b2, 1e-8,
                                            1 - b1 ** step, 1 - b2 ** step, sp()), 'adam')
        gm = gs.double().clone()
        gm[mask > 0] = 0
        pr[(mask & 2) > 0] = 0
        pr, mr, vr = adam_step_ref(pr, gm, mr, vr, step, lr, b1, b2)
    assert rel_err(pd, pr) < 1e-5
    e = synth_tensor('opt/e', (n,))
    ed = e.to(DEV)
    _lib.check(lib.rick_ema_f32(ptr(ed), ptr(pd), n, 0.997784, sp()), 'ema')
    assert rel_err(ed, e.double() * 0.997784 + pd.double().cpu() * (1 - 0.997784)) < 1e-6


@pytest.mark.parametrize('B,O,I,k', [(4, 512, 512, 3), (3, 24, 16, 3), (1, 128, 256, 3), (8, 64, 36, 1), (32, 8, 8, 3)])
def test_demod_fused_matches_composed(B, O, I, k):
    """rick_wsq / rick_demod / rick_demod_bwd_*: values and first-order gradients of the fused demodulation equal the
    tensor-algebra form (model_probe_tune.py:246-252) evaluated in fp64."""
    from rick_amd.op.modconv import demod_coeff, demod_coeff_fused
    gen = torch.Generator().manual_seed(B * 1000 + O + I)
    w = torch.randn(O, I, k, k, generator=gen)
    s = torch.randn(B, I, generator=gen) * 0.7 + 1.0
    gd = torch.randn(B, O, generator=gen)
    scale = 1.0 / (I * k * k) ** 0.5
    w64, s64 = w.double().requires_grad_(True), s.double().requires_grad_(True)
    d_ref = demod_coeff(w64, s64, scale)
    gw_ref, gs_ref = torch.autograd.grad(d_ref, [w64, s64], gd.double())
    wd, sd = w.to(DEV).requires_grad_(True), s.to(DEV).requires_grad_(True)
    d = demod_coeff_fused(wd, sd, scale)
    gw, gs = torch.autograd.grad(d, [wd, sd], gd.to(DEV))
    assert rel_err(d, d_ref.detach()) < 2e-6
    assert rel_err(gs, gs_ref) < 5e-6
    assert rel_err(gw, gw_ref) < 5e-6
    # only one of the two gradients requested
    d2 = demod_coeff_fused(wd.detach(), sd, scale)
    (gs2,) = torch.autograd.grad(d2, [sd], gd.to(DEV))
    assert torch.equal(gs2, gs)


@pytest.mark.parametrize('N,I,O,H,k,s,p', [(2, 16, 24, 9, 3, 1, 1), (4, 512, 512, 8, 3, 1, 1), (3, 32, 64, 17, 3, 2, 0),
                                           (2, 64, 128, 16, 1, 1, 0), (8, 128, 128, 32, 3, 1, 1)])
def test_conv_bias_act_fused_equals_two_pass(N, I, O, H, k, s, p):
    """rick_conv_igemm_act_f32: the fused tail is bit-identical to conv2d -> fused_leaky_relu (also through the
    split-K second stage), and so are the input / weight / bias gradients."""
    from rick_amd import op
    gen = torch.Generator().manual_seed(N * 100 + I + O + H)
    x = torch.randn(N, I, H, H, generator=gen).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(O, I, k, k, generator=gen).to(DEV).requires_grad_(True)
    b = torch.randn(O, generator=gen).to(DEV).requires_grad_(True)
    scale = 1.0 / (I * k * k) ** 0.5
    y1 = op.fused_leaky_relu(op.conv2d(x, w, s, p, wscale=scale), b)
    y2 = op.conv2d_bias_act(x, w, b, s, p, wscale=scale)
    assert torch.equal(y1, y2)
    g = torch.randn(y1.shape, generator=gen).to(DEV)
    g1 = torch.autograd.grad(y1, [x, w, b], g)
    g2 = torch.autograd.grad(y2, [x, w, b], g)
    for a, c in zip(g1, g2):
        assert torch.equal(a, c)


@pytest.mark.parametrize('N,C,H,pad,nb', [(2, 64, 9, (1, 1), 2), (4, 128, 33, (1, 1), 1), (3, 512, 17, (2, 2), 3), (2, 24, 9, (1, 1), 2)])
def test_blur_noise_bias_act_fused_equals_two_pass(N, C, H, pad, nb):
    """rick_upfirdn2d_act_f32 (blur + NoiseInjection + bias + LeakyReLU in one launch) is bit-identical to
    upfirdn2d -> fused_noise_bias_act, forward and backward; C = 24 takes the documented two-op fallback."""
    from rick_amd import op
    gen = torch.Generator().manual_seed(N * 100 + C + H)
    k = torch.tensor([1., 3., 3., 1.])
    k = (torch.outer(k, k) / 64 * 4).to(DEV)
    x = torch.randn(N, C, H, H, generator=gen).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = torch.randn(C, generator=gen).to(DEV).requires_grad_(True)
    nw = torch.tensor([0.3], device=DEV, requires_grad=True)
    oh = H + 2 * pad[0] - 3
    noise = torch.randn(nb if nb == 1 else N, 1, oh, oh, generator=gen).to(DEV)
    y1 = op.fused_noise_bias_act(op.upfirdn2d(x, k, pad=pad), b, noise, nw)
    y2 = op.upfirdn2d_noise_bias_act(x, k, pad, b, noise, nw)
    assert torch.equal(y1, y2)
    g = torch.randn(y1.shape, generator=gen).to(DEV)
    for a, c in zip(torch.autograd.grad(y1, [x, b, nw], g), torch.autograd.grad(y2, [x, b, nw], g)):
        assert torch.equal(a, c)


@pytest.mark.parametrize('B,I,O,H,nb', [(2, 16, 24, 9, 2), (4, 512, 512, 8, 1), (3, 64, 128, 32, 3), (4, 128, 128, 64, 4)])
def test_modconv_fused_tail_equals_two_pass(B, I, O, H, nb):
    """Plain StyledConv with NoiseInjection + bias + LeakyReLU in the conv epilogue: forward bit-identical to
    modulated_conv_fused -> fused_noise_bias_act; gradients identical except the demodulation gradient, which is
    rebuilt from the saved activation (rick_hw_dot_act_f32) and agrees to rounding."""
    from rick_amd import op
    from rick_amd.op.modconv import modulated_conv_fused
    gen = torch.Generator().manual_seed(B * 100 + I + O + H)
    mk = lambda *s: torch.randn(*s, generator=gen).to(DEV)         # noqa: E731
    x = mk(B, I, H, H).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = mk(O, I, 3, 3).requires_grad_(True)
    s = (mk(B, I) * 0.5 + 1).requires_grad_(True)
    d = (torch.rand(B, O, generator=gen).to(DEV) + 0.5).requires_grad_(True)
    b = mk(O).requires_grad_(True)
    nw = torch.tensor([0.4], device=DEV, requires_grad=True)
    noise = mk(nb if nb == 1 else B, 1, H, H)
    scale = 1.0 / (I * 9) ** 0.5
    y1 = op.fused_noise_bias_act(modulated_conv_fused(x, w, s, d, scale, False), b, noise, nw)
    y2 = modulated_conv_fused(x, w, s, d, scale, False, None, (b, noise, nw, 0.2, 2 ** 0.5))
    assert torch.equal(y1, y2)
    g = mk(*y1.shape)
    g1 = torch.autograd.grad(y1, [x, w, s, d, b, nw], g)
    g2 = torch.autograd.grad(y2, [x, w, s, d, b, nw], g)
    for k, (a, c) in enumerate(zip(g1, g2)):
        if k == 3:
            assert rel_err(c, a) < 2e-5
        else:
            assert torch.equal(a, c), k


@pytest.mark.parametrize('B', [1, 2, 4, 8])
def test_modulation_bank_equals_per_layer_linears(B):
    """rick_modbank_{fwd,bwd}_f32 (every modulation EqualLinear of a generator in one launch) vs the per-layer
    EqualLinear path (model_probe_tune.py:139-173,233): values, weight and bias gradients."""
    from rick_amd.models import Generator
    torch.manual_seed(5)
    g = Generator(64, 512, 8).to(DEV)
    bank = g._modulation_bank()
    for m in bank.linears:                         # non-trivial biases
        m.bias.data.normal_()
    lat = torch.randn(B, g.n_latent, 512, device=DEV)
    outs = bank(lat)
    refs = [m(lat[:, i]) for m, i in zip(bank.linears, bank.lat_idx)]
    assert len(outs) == len(refs) == 2 + 3 * len(g.to_rgbs)
    for o, r in zip(outs, refs):
        assert o.shape == r.shape and o.is_contiguous()
        assert rel_err(o, r.double()) < 2e-6
    gs = [torch.randn_like(o) for o in outs]
    params = bank.params()
    got = torch.autograd.grad(outs, params, gs)
    ref = torch.autograd.grad(refs, params, gs)
    for a, b in zip(got, ref):
        assert rel_err(a, b.double()) < 3e-6
    # unused outputs (None gradients) are treated as zeros
    got2 = torch.autograd.grad(bank(lat)[3].sum(), [params[6], params[7]], allow_unused=True)
    assert rel_err(got2[1], torch.full_like(got2[1], float(B)).double()) < 1e-6


def test_unused_reference_variants_downsample_modconv_and_scaled_lrelu():
    """ModulatedConv2d(downsample=True) (model_probe_tune.py:214-220, 270-276) and the ScaledLeakyReLU branch of ConvLayer
    (:176-185, 635-639) — never instantiated by the reference networks, present for API completeness — against the
    reference's own formulation restated in fp64 (per-sample weights, grouped conv)."""
    from oracle.ops_ref import make_blur_kernel, upfirdn2d_ref
    from rick_amd.models import ConvLayer, ModulatedConv2d, ScaledLeakyReLU
    B, CI, CO, R, SD = 2, 16, 24, 9, 32
    m = ModulatedConv2d(CI, CO, 3, SD, downsample=True)
    m.load_state_dict(synth_state_dict({k: v.shape for k, v in m.state_dict().items()}), strict=False)
    x = synth_tensor('down/x', (B, CI, R, R))
    st = synth_tensor('down/s', (B, SD))
    y = m.to(DEV)(x.to(DEV), st.to(DEV))
    sd = {k: v.double().cpu() for k, v in m.state_dict().items()}
    s = st.double() @ (sd['modulation.weight'] * (1 / math.sqrt(SD))).t() + sd['modulation.bias']
    w = m.scale * sd['weight'] * s.view(B, 1, CI, 1, 1)
    w = w * torch.rsqrt(w.pow(2).sum([2, 3, 4]) + 1e-8).view(B, CO, 1, 1, 1)
    p = (4 - 2) + (3 - 1)
    xb = upfirdn2d_ref(x.double(), make_blur_kernel([1, 3, 3, 1]).double(), pad=((p + 1) // 2, p // 2))
    ref = F.conv2d(xb.reshape(1, B * CI, *xb.shape[2:]), w.reshape(B * CO, CI, 3, 3), stride=2, groups=B)
    ref = ref.view(B, CO, *ref.shape[2:])
    assert y.shape == ref.shape and rel_err(y, ref) < 3e-5
    layer = ConvLayer(8, 12, 3, bias=False, activate=True).to(DEV)
    assert isinstance(layer[-1], ScaledLeakyReLU) and layer[0].bias is None
    xi = synth_tensor('slr/x', (2, 8, 6, 6))
    out = layer(xi.to(DEV))
    refo = F.leaky_relu(F.conv2d(xi.double(), layer[0].weight.detach().double().cpu() * layer[0].scale, padding=1), 0.2) * math.sqrt(2)
    assert rel_err(out, refo) < 3e-5


def test_gradient_sink_equals_autograd_accumulation():
    """op.grad_sink(): the bias / noise-strength sums of the activation adjoint (rick_bias_act_bwd_f32, accumulate), the
    conv weight gradients and the ModulationBank gradients are ADDED straight into the parameters' .grad — the values
    must equal what autograd's AccumulateGrad leaves there (same fp32 additions), starting from a non-zero .grad."""
    from rick_amd import op
    from rick_amd.models import Generator
    from rick_amd.op.modconv import modulated_conv_fused
    gen = torch.Generator().manual_seed(11)
    mk = lambda *s: torch.randn(*s, generator=gen).to(DEV)         # noqa: E731
    B, I, O, H = 2, 64, 64, 16
    params = {'w1': mk(O, I, 3, 3), 'b1': mk(O), 'w2': mk(O, O, 3, 3), 'b2': mk(O), 'nw2': mk(1), 'b3': mk(O), 'nw3': mk(1), 'b4': mk(O)}
    x = mk(B, I, H, H).contiguous(memory_format=torch.channels_last)
    s, d = mk(B, O) * 0.5 + 1, torch.rand(B, O, generator=gen).to(DEV) + 0.5
    noise = mk(B, 1, H, H)
    k = torch.tensor([1., 3., 3., 1.])
    k = (torch.outer(k, k) / 64).to(DEV)
    init = {n: mk(*p.shape) for n, p in params.items()}
    gout = mk(B, O, H, H)

    def run(sink):
        ps = {n: p.clone().requires_grad_(True) for n, p in params.items()}
        for n, p in ps.items():
            p.grad = init[n].clone()
        ctx = op.grad_sink() if sink else contextlib.nullcontext()
        with ctx:
            y = op.conv2d_bias_act(x, ps['w1'], ps['b1'], 1, 1, wscale=0.05, key=(ps['w1'], 'w'))
            y = modulated_conv_fused(y, ps['w2'], s, d, 0.04, False, (ps['w2'], 'mod'), (ps['b2'], noise, ps['nw2'], 0.2, 2 ** 0.5))
            y = op.upfirdn2d_noise_bias_act(y, k * 4, (2, 1), ps['b3'], noise, ps['nw3'])
            y = op.fused_leaky_relu(y, ps['b4'])
            y.backward(gout)
        return {n: p.grad.clone() for n, p in ps.items()}

    ref, got = run(False), run(True)
    for n in ref:
        assert torch.equal(ref[n], got[n]), n
        assert not torch.equal(ref[n], init[n]), n
    # ModulationBank: weight / bias gradients of the trainable linears land in .grad, frozen layers are skipped
    torch.manual_seed(3)
    g = Generator(32, 512, 2).to(DEV)
    bank = g._modulation_bank()
    for m in bank.linears[1::3]:                       # freeze the ToRGB modulations (as the trainer does)
        m.weight.requires_grad = m.bias.requires_grad = False
    lat = torch.randn(2, g.n_latent, 512, device=DEV)
    gs = None
    res = []
    for sink in (False, True):
        for p in bank.params():
            p.grad = torch.full_like(p, 0.25) if p.requires_grad else None
        with (op.grad_sink() if sink else contextlib.nullcontext()):
            outs = [o for o in bank(lat) if o.requires_grad]
            gs = gs or [torch.randn_like(o) for o in outs]
            torch.autograd.backward(outs, gs)
        res.append([p.grad.clone() if p.grad is not None else None for p in bank.params()])
    for a, b in zip(*res):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)


@pytest.mark.parametrize('B,K,O,act,pn', [(8, 512, 512, True, True), (1, 512, 512, True, False), (16, 512, 256, False, False),
                                          (3, 64, 40, True, True)])
def test_equal_linear_short_batch_kernel(B, K, O, act, pn):
    """rick_equal_linear_f32 ([PixelNorm +] EqualLinear + bias * lr_mul + fused activation in one launch,
    model_probe_tune.py:92-98, 139-173) vs the same formula in fp64."""
    from rick_amd import op
    gen = torch.Generator().manual_seed(B + K + O)
    x = torch.randn(B, K, generator=gen).to(DEV)
    w = (torch.randn(O, K, generator=gen) / 0.01).to(DEV)
    b = torch.randn(O, generator=gen).to(DEV)
    scale, lr_mul = (1 / math.sqrt(K)) * 0.01, 0.01
    y = op.equal_linear(x, w, b, scale, lr_mul, act, pn)
    xd = x.double()
    if pn:
        xd = xd * torch.rsqrt(xd.pow(2).mean(1, keepdim=True) + 1e-8)
    ref = xd @ (w.double() * scale).t() + b.double() * lr_mul
    if act:
        ref = F.leaky_relu(ref, 0.2) * math.sqrt(2)
    assert y.shape == ref.shape and rel_err(y, ref) < 2e-6


@pytest.mark.parametrize('B,K,O,bias', [(8, 8192, 512, False), (4, 8192, 512, True), (8, 512, 1, True), (1, 512, 512, True),
                                        (16, 2052, 37, True), (3, 64, 40, False)])
def test_linear_family_values_and_gradients_vs_fp64(B, K, O, bias):
    """op.linear = F.linear(x, W * scale, bias * lr_mul) (model_probe_tune.py:157-168) on rick_linear_{fwd,dgrad,wgrad}_f32:
    value, the three first-order gradients and the second-order terms R1 needs (gradient of |dy/dx|^2 w.r.t. W, b and x
    through the family's own double backward) against the same expressions in float64."""
    from rick_amd import op
    gen = torch.Generator().manual_seed(B * 7 + K + O)
    x = torch.randn(B, K, generator=gen).to(DEV).requires_grad_(True)
    w = torch.randn(O, K, generator=gen).to(DEV).requires_grad_(True)
    b = torch.randn(O, generator=gen).to(DEV).requires_grad_(True) if bias else None
    gy = torch.randn(B, O, generator=gen).to(DEV)
    alpha, mul = 1 / math.sqrt(K), 0.5
    xd, wd, gyd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True), gy.double()
    bd = b.detach().double().requires_grad_(True) if bias else None

    def f32():
        return op.linear(x, w, b, alpha, mul)

    def f64():
        return xd @ (wd * alpha).t() + (bd * mul if bias else 0.0)
    y, yd = f32(), f64()
    assert rel_err(y, yd) < 2e-6
    ins, insd = [t for t in (x, w, b) if t is not None], [t for t in (xd, wd, bd) if t is not None]
    g1 = torch.autograd.grad(y, ins, gy)
    g1d = torch.autograd.grad(yd, insd, gyd)
    for a, r in zip(g1, g1d):
        assert rel_err(a, r) < 2e-6
    # second order: h = d<y, gy>/dx (depends on W only), loss = sum (h * c)^2 + sum y  -> gradients w.r.t. W, b (and x through y)
    c = torch.randn(B, K, generator=gen).to(DEV)
    (h,) = torch.autograd.grad(f32(), x, gy, create_graph=True)
    (hd,) = torch.autograd.grad(f64(), xd, gyd, create_graph=True)
    assert rel_err(h, hd) < 2e-6
    g2 = torch.autograd.grad((h * c).pow(2).sum() + (f32() * gy).sum(), ins)
    g2d = torch.autograd.grad((hd * c.double()).pow(2).sum() + (f64() * gyd).sum(), insd)
    for a, r in zip(g2, g2d):
        assert rel_err(a, r) < 5e-6
    # third member's double backward: gradient of a function of dL/dW w.r.t. the upstream gradient and x
    gyr = gy.clone().requires_grad_(True)
    gyrd = gyd.clone().requires_grad_(True)
    (gw,) = torch.autograd.grad(f32(), w, gyr, create_graph=True)
    (gwd,) = torch.autograd.grad(f64(), wd, gyrd, create_graph=True)
    g3 = torch.autograd.grad(gw.pow(2).sum(), [gyr, x])
    g3d = torch.autograd.grad(gwd.pow(2).sum(), [gyrd, xd])
    for a, r in zip(g3, g3d):
        assert rel_err(a, r) < 5e-6


def test_linear_family_is_bit_reproducible_and_sinks_add_in_place():
    """Fixed summation order: 200 evaluations of forward / data gradient / weight gradient of the 8192 -> 512 layer give ONE
    bit pattern each.  Under op.grad_sink() the weight and bias gradients are added into the parameters' .grad by the kernel
    (nothing returned to autograd), equal to .grad + gradient of the plain path."""
    from rick_amd import op
    import importlib
    L = importlib.import_module('rick_amd.op.linear')      # (the package attribute `linear` is the function)
    torch.manual_seed(4)
    B, K, O = 8, 8192, 512
    x, w, g = torch.randn(B, K, device=DEV), torch.randn(O, K, device=DEV), torch.randn(B, O, device=DEV)
    first = (L._p1(x, w, None, 0.01, 0.0), L._p2(g, w, 0.01), L._p3(g, x, 0.01)[0])
    for _ in range(200):
        again = (L._p1(x, w, None, 0.01, 0.0), L._p2(g, w, 0.01), L._p3(g, x, 0.01)[0])
        assert all(torch.equal(a, b) for a, b in zip(first, again))
    wp, bp = torch.nn.Parameter(w.clone()), torch.nn.Parameter(torch.randn(O, device=DEV))
    xr = x.clone().requires_grad_(True)
    (op.linear(xr, wp, bp, 0.01, 0.5) * g).sum().backward()
    gw0, gb0, gx0 = wp.grad.clone(), bp.grad.clone(), xr.grad.clone()
    base_w, base_b = torch.randn_like(wp), torch.randn_like(bp)
    wp.grad, bp.grad, xr.grad = base_w.clone(), base_b.clone(), None
    with op.grad_sink():
        (op.linear(xr, wp, bp, 0.01, 0.5) * g).sum().backward()
    assert torch.equal(xr.grad, gx0)
    assert torch.equal(wp.grad, base_w + gw0) and torch.equal(bp.grad, base_b + gb0)


def test_discriminator_final_layers_run_the_linear_family():
    """EqualLinear with gradients at batch <= 16 (D's final_linear, model_probe_tune.py:699-702) runs op.linear: same logits and
    gradients as the BLAS route (RICK_NO_OWN_LINEAR) to fp32 rounding, and no GEMM launch from torch in a D forward + backward."""
    import rick_amd.models as M
    torch.manual_seed(6)
    d = M.Discriminator(32).to(DEV)
    img = torch.randn(4, 3, 32, 32, device=DEV)
    res = {}
    for own in (True, False):
        M._USE_OWN_LINEAR = own
        try:
            d.zero_grad(set_to_none=True)
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                out, _ = d(img)
                out.square().sum().backward()
                torch.cuda.synchronize()
            gemms = [ev.key for ev in prof.key_averages() if 'Cijk' in ev.key or 'rocblas' in ev.key.lower() or 'hipblas' in ev.key.lower()]
            res[own] = (out.detach().clone(), {n: p.grad.clone() for n, p in d.named_parameters() if 'final_linear' in n}, gemms)
        finally:
            M._USE_OWN_LINEAR = True
    assert not res[True][2], res[True][2]
    assert res[False][2]                                        # (the switch does switch)
    assert rel_err(res[True][0], res[False][0]) < 1e-5
    for k in res[True][1]:
        assert rel_err(res[True][1][k], res[False][1][k]) < 1e-5, k


@pytest.mark.parametrize('n,c,r,sink', [(4, 512, 16, True), (2, 128, 64, False), (4, 512, 4, True), (1, 64, 32, False)])
def test_activation_adjoint_with_fused_demodulation_gradient(n, c, r, sink):
    """rick_bias_act_bwd_dot_f32: the activation adjoint of a generator layer and, from the same pass over (g, y), the layer's
    demodulation gradient gd = sum_hw adjoint * conv_out / d — gx / gb / gnw bit-identical to the plain adjoint
    (rick_bias_act_bwd_f32), gd equal to the two-pass route (rick_hw_dot_act_f32) to rounding and to the fp64 formula."""
    from rick_amd.op import fused_act as fa
    from rick_amd.op.modconv import _hw_dot_act_raw
    gen = torch.Generator().manual_seed(n * 1000 + c + r)
    def nhwc(*shape):
        return torch.randn(*shape, generator=gen).to(DEV).contiguous(memory_format=torch.channels_last)
    x, g = nhwc(n, c, r, r), nhwc(n, c, r, r)
    bias = torch.randn(c, generator=gen).to(DEV)
    noise = torch.randn(1, 1, r, r, generator=gen).to(DEV)
    nw = torch.full((1,), 0.3, device=DEV)
    d = (torch.rand(n, c, generator=gen) + 0.5).to(DEV)
    slope, gain = 0.2, math.sqrt(2)
    y = fa.fused_noise_bias_act(x, bias, noise, nw, slope, gain)
    sb, sw = (torch.randn(c, device=DEV), torch.randn(1, device=DEV)) if sink else (None, None)
    sb0, sw0 = (sb.clone(), sw.clone()) if sink else (None, None)
    gx0, gb0, gnw0 = fa._ActAdjoint.apply(g, y, noise, slope, gain, True, True, sb0, sw0)
    gd0 = _hw_dot_act_raw(gx0, y, bias, noise, nw, slope, gain, divisor=d)
    res = fa.act_adjoint_dot(g, y, noise, slope, gain, True, True, sb, sw, bias, nw, d)
    assert res is not None
    gx, gb, gnw, gd = res
    assert torch.equal(gx, gx0)
    if sink:
        assert gb is None and gnw is None and torch.equal(sb, sb0) and torch.equal(sw, sw0)
    else:
        assert torch.equal(gb, gb0) and torch.equal(gnw, gnw0)
    assert rel_err(gd, gd0) < 2e-6
    # fp64: conv_out = x (the pre-activation without noise and bias), gd = sum_hw gx * x / d
    ref = (gx0.double() * x.double()).sum((2, 3)) / d.double()
    assert rel_err(gd, ref) < 1e-5
    # a frozen layer (conv1 of the generator under the G optimiser): no parameter gradient wanted, gd still is
    gx2, gb2, gnw2, gd2 = fa.act_adjoint_dot(g, y, noise, slope, gain, False, False, None, None, bias, nw, d)
    assert gb2 is None and gnw2 is None and torch.equal(gx2, gx0) and torch.equal(gd2, gd)


def test_mapping_network_fast_path_equals_autograd_path():
    """Generator.style without autograd (the train steps' latents) runs the one-launch-per-layer kernels; with autograd
    (Fisher sweep, tests differentiating through the mapping network) the rocBLAS + fused-activation path.  Same values."""
    from rick_amd.models import Generator
    torch.manual_seed(9)
    g = Generator(32, 512, 8).to(DEV)
    z = torch.randn(8, 512, device=DEV)
    with torch.no_grad():
        fast = g.style(z)
    slow = g.style(z.clone().requires_grad_(True))
    assert slow.requires_grad and not fast.requires_grad
    assert rel_err(fast, slow.detach().double()) < 5e-6
    # randomize_noise draws every layer's noise map from one launch: right shapes, fresh values per call
    for n, p in g.named_parameters():
        if n.endswith('noise.weight'):
            p.data.fill_(0.5)
    with torch.no_grad():
        a, _ = g([z[:2]])
        b, _ = g([z[:2]])
    assert a.shape == (2, 3, 32, 32) and torch.isfinite(a).all() and not torch.equal(a, b)


@pytest.mark.parametrize('B,C,H,skip', [(2, 512, 8, False), (4, 128, 64, True), (3, 40, 17, True)])
def test_torgb_fused_equals_composed(B, C, H, skip):
    """rick_torgb_{fwd,bwdx}_f32 (weight modulation + bias + skip addition inside the thin product's launch) vs the
    composed ToRGB (per-sample weight tensor, thin_fwd, two adds: the path second-order autograd keeps): identical
    values and data gradient, parameter / style gradients to rounding."""
    from rick_amd import op
    from rick_amd.models import ToRGB
    torch.manual_seed(B + C + H)
    m = ToRGB(C, 64, upsample=skip).to(DEV)
    m.bias.data.normal_()
    x = torch.randn(B, C, H, H, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    st = torch.randn(B, 64, device=DEV, requires_grad=True)
    sk = torch.randn(B, 3, H // 2, H // 2, device=DEV, requires_grad=True) if skip and H % 2 == 0 else None
    if skip and sk is None:
        sk = torch.randn(B, 3, (H + 1) // 2, (H + 1) // 2, device=DEV, requires_grad=True)
        x = torch.randn(B, C, 2 * sk.shape[2], 2 * sk.shape[2], device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ins = [x, st] + ([sk] if sk is not None else []) + [m.conv.weight, m.bias, m.conv.modulation.weight]
    y1 = m(x, st, sk)
    with op.second_order():
        y2 = m(x, st, sk)
    assert torch.equal(y1, y2)
    g = torch.randn_like(y1)
    g1 = torch.autograd.grad(y1, ins, g)
    g2 = torch.autograd.grad(y2, ins, g)
    assert torch.equal(g1[0], g2[0])
    for a, b in zip(g1[1:], g2[1:]):
        assert rel_err(a, b.double()) < 1e-5


@pytest.mark.parametrize('O,I,k', [(512, 512, 3), (128, 256, 3), (24, 16, 3), (130, 70, 3), (256, 128, 1), (40, 24, 1)])
def test_group_pack_equals_single_pack(O, I, k):
    """rick_conv_pack_weights_multi (one launch per network; source read through LDS in its own memory order) writes
    the same bytes as the per-weight pack kernel — for the parameter's own layout, its transposed view (data-gradient
    operand) and ragged channel counts."""
    from rick_amd.op import conv as cv
    torch.manual_seed(O + I + k)
    lin = torch.nn.Linear(1, 1)                      # a module to hang the parameter on (PackGroup is per network)
    lin.w = torch.nn.Parameter(torch.randn(O, I, k, k, device=DEV))
    grp = cv.register_pack_group(lin)
    for view, tag in ((lin.w, 'a'), (lin.w.transpose(0, 1), 'b')):
        first = cv._pack(view, 0.37, (lin.w, tag)).clone()      # registration packs this one with the single kernel
        lin.w.data.mul_(1.5)
        cv.bump_weights_epoch([lin.w])
        assert grp.refresh()                                     # group launch over every registered request
        multi = cv._pack(view, 0.37, (lin.w, tag)).clone()
        single = cv._pack(view, 0.37, None)
        assert torch.equal(multi, single) and not torch.equal(multi, first)


@pytest.mark.parametrize('case', ['tiny', 'huge', 'zero_first_chunk', 'pruned_and_tiny', 'range'])
def test_conv_operand_exponent_edge_cases(case):
    """The fp16 hi/lo split multiplies each operand by a power of two taken from a sample of the block's own data
    (conv_common.h).  Results must stay fp32-grade when the data sit far from 1 (gradients: 1e-8; large activations),
    when the sampled first channel chunk is entirely zero (pruned filters: the fallback sample over all chunks) and
    when most values are 2^-10 of the maximum (lo parts are fp16 subnormals)."""
    from rick_amd import op
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 96, 16, 16, generator=g)
    w = torch.randn(128, 96, 3, 3, generator=g)
    tol = 2e-6
    if case == 'tiny':
        x, w = x * 1e-8, w * 1e-5
    elif case == 'huge':
        x = x * 3e4
    elif case == 'zero_first_chunk':
        x[:, :32] = 0
    elif case == 'pruned_and_tiny':
        x = x * 1e-9
        x[:, :32] = 0
        w[5] = 0
        w[:, 40] = 0
    elif case == 'range':
        x = x * 2.0 ** -10
        x[:, 0] = torch.randn(2, 16, 16, generator=g)
        w[:, 0] = 0
        tol = 2e-5          # lo parts of the small values are fp16 subnormals: absolute error 2^-27 of the block maximum
    wscale = 1 / math.sqrt(96 * 9)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr * wscale, padding=1)
    gy = torch.randn(yr.shape, generator=g) * float(yr.abs().max())
    gxr, gwr = torch.autograd.grad(yr, (xr, wr), gy.double())
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = op.conv2d(xd, wd, 1, 1, wscale=wscale)
    gx, gw = torch.autograd.grad(y, (xd, wd), gy.to(DEV))
    assert torch.isfinite(y).all()
    assert rel_err(y, yr) < tol, rel_err(y, yr)
    assert rel_err(gx, gxr) < 2e-6, rel_err(gx, gxr)
    assert rel_err(gw, gwr) < 2e-6, rel_err(gw, gwr)
    # transposed stride-2 form (its own kernel)
    yt = op.conv_transpose2d(xd, wd[:64], 2, 0, wscale=wscale)                 # weight given as [O, I, kh, kw]
    ytr = F.conv_transpose2d(xr, (wr[:64] * wscale).transpose(0, 1), stride=2)  # torch wants [I, O, kh, kw]
    assert rel_err(yt, ytr) < tol, rel_err(yt, ytr)


# ------------------------------------------------------------------ the extensions' other dtypes (half / double)
@pytest.mark.parametrize('up,down,pad', [(1, 1, (2, 1)), (2, 1, (2, 1)), (1, 2, (1, 1)), (2, 3, (0, 2))])
def test_upfirdn2d_double_vs_oracle_and_gradcheck(up, down, pad):
    """op.upfirdn2d in float64 (op/upfirdn2d_kernel.cu:311-367 dispatches half / float / double; the reference's gradcheck
    idiom is `.double()`): values against the oracle at 1e-13, first and second derivatives by torch's own gradcheck /
    gradgradcheck against finite differences."""
    from oracle.ops_ref import upfirdn2d_ref
    from rick_amd import op
    torch.manual_seed(0)
    k = torch.tensor([1., 3., 3., 1.], dtype=torch.float64)
    k = (k[:, None] * k[None, :] / k.sum() ** 2).to(DEV)
    x = torch.randn(2, 3, 9, 7, dtype=torch.float64, device=DEV)
    y = op.upfirdn2d(x, k, up=up, down=down, pad=pad)
    ref = upfirdn2d_ref(x.cpu(), k.cpu(), up=up, down=down, pad=pad)
    assert y.dtype == torch.float64 and y.shape == ref.shape
    assert float((y.cpu() - ref).abs().max()) < 1e-13
    xs = torch.randn(1, 2, 5, 4, dtype=torch.float64, device=DEV, requires_grad=True)
    f = lambda t: op.upfirdn2d(t, k, up=up, down=down, pad=pad)   # noqa: E731
    assert torch.autograd.gradcheck(f, (xs,), eps=1e-6, atol=1e-8)
    assert torch.autograd.gradgradcheck(f, (xs,), eps=1e-6, atol=1e-8)


def test_fused_leaky_relu_double_gradcheck_and_half():
    """op.fused_leaky_relu in float64 (values vs the oracle at 1e-15, gradcheck / gradgradcheck incl. the bias) and float16
    (against the float32 op at half precision); mixed dtypes are refused like the extension's TORCH_CHECK."""
    from oracle.ops_ref import fused_leaky_relu_ref
    from rick_amd import op
    torch.manual_seed(1)
    x = torch.randn(3, 8, 5, 4, dtype=torch.float64, device=DEV)
    b = torch.randn(8, dtype=torch.float64, device=DEV)
    y = op.fused_leaky_relu(x, b, 0.2, 2 ** 0.5)
    # (alpha and scale cross the binding as C floats, in the reference too: op/fused_bias_act.cpp:11-13)
    a32, s32 = float(np.float32(0.2)), float(np.float32(2 ** 0.5))
    assert float((y.cpu() - fused_leaky_relu_ref(x.cpu(), b.cpu(), a32, s32)).abs().max()) < 1e-14
    # keep away from the kink for the finite differences
    xs = (torch.randn(2, 4, 3, 3, dtype=torch.float64, device=DEV).sign() * (0.2 + torch.rand(2, 4, 3, 3, dtype=torch.float64, device=DEV))).requires_grad_(True)
    bs = torch.zeros(4, dtype=torch.float64, device=DEV, requires_grad=True)
    f = lambda t, u: op.fused_leaky_relu(t, u, 0.2, 2 ** 0.5)   # noqa: E731
    assert torch.autograd.gradcheck(f, (xs, bs), eps=1e-6, atol=1e-8)
    assert torch.autograd.gradgradcheck(f, (xs, bs), eps=1e-6, atol=1e-8)
    xh, bh = x.half(), b.half()
    yh = op.fused_leaky_relu(xh, bh, 0.2, 2 ** 0.5)
    y32 = op.fused_leaky_relu(xh.float(), bh.float(), 0.2, 2 ** 0.5)
    assert yh.dtype == torch.float16 and float((yh.float() - y32).abs().max()) <= 2e-3 * float(y32.abs().max())
    kh = torch.ones(4, 4, dtype=torch.float16, device=DEV) / 16
    uh = op.upfirdn2d(xh, kh, pad=(2, 1))
    u32 = op.upfirdn2d(xh.float(), kh.float(), pad=(2, 1))
    assert uh.dtype == torch.float16 and float((uh.float() - u32).abs().max()) <= 2e-3 * float(u32.abs().max())
    with pytest.raises(RuntimeError):
        op.fused_leaky_relu(x, b.float())
    with pytest.raises(RuntimeError):
        op.upfirdn2d(x, kh)
