"""Split images (rick_amd/op/split.py, csrc/split.hip): the pre-split activation format of the MFMA kernels, its producers
(FIR, activation adjoint, ResBlock merge, conv epilogue maxima), the split-operand forms of igemm / convt2 / wgrad, and the
one-node discriminator ResBlock built on them (rick_amd/op/dblock.py) against the per-layer path and fp64 references."""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _sat(reset=False):
    from rick_amd._lib import check, lib
    c = ctypes.c_uint(0)
    check(lib.rick_saturation_count(ctypes.byref(c), int(reset)), 'rick_saturation_count')
    return c.value


def _octaves(shape, lo=-8, hi=3, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(shape, generator=g) * torch.exp2(torch.randint(lo, hi, (shape[0], shape[1], 1, 1), generator=g).float())
    return _cl(x.to(DEV))


def test_amax_and_pack_roundtrip():
    from rick_amd.op import split as sp
    _sat(reset=True)
    for shape in [(2, 64, 9, 7), (1, 32, 1, 1), (3, 128, 17, 5)]:
        x = _octaves(shape, seed=shape[1])
        a = sp.amax(x)
        assert a.numel() == sp.AMAX_FLOATS and float(sp.amax_value(a)) == float(x.abs().max())
        img = sp.split_pack(x)
        scale, unscale, bound = [float(v) for v in img.hdr[:3]]
        assert bound == float(x.abs().max()) and scale * unscale == 1.0 and 2 ** 13 <= bound * scale < 2 ** 14
        y = sp.split_unpack(img)
        # |x - hi - lo| <= 2^-22 |x| for values within 2^10 of the bound; absolute 2^-25 / 2^13 of the bound below
        tol = torch.maximum(x.abs() * 2.0 ** -21, torch.full_like(x, bound * 2.0 ** -37))
        assert bool(((y - x).abs() <= tol).all())
        # the image is exactly {fp16(x * 2^e) x 4 | fp16(x * 2^e - hi) x 4} in the 16 bytes of every 4 channels
        n, c, h, w = shape
        raw = img.data.permute(0, 2, 3, 1).contiguous().view(torch.float16).view(n, h, w, c // 4, 2, 4)
        xs = (x.permute(0, 2, 3, 1) * scale).view(n, h, w, c // 4, 4)
        hi = xs.half()
        lo = (xs - hi.float()).half()
        assert torch.equal(raw[..., 0, :], hi) and torch.equal(raw[..., 1, :], lo)
    assert _sat() == 0


@pytest.mark.saturates
def test_wrong_bound_is_counted_not_silent():
    """A producer whose values exceed the bound it was given (a caller's mistake: bounds are guaranteed, not sampled) is
    counted AND loud: the producers do not set MODE.FP16_OVFL, so the value becomes inf / NaN instead of a finite wrong one."""
    from rick_amd.op import split as sp
    _sat(reset=True)
    x = _cl(torch.randn(1, 32, 8, 8, device=DEV))
    x[0, 3, 2, 2] = 1e4
    lie = torch.full((sp.AMAX_FLOATS,), 1.0, device=DEV)
    img = sp.split_pack(x, lie)
    torch.cuda.synchronize()
    assert _sat() > 0
    y = sp.split_unpack(img)
    assert not bool(torch.isfinite(y[0, 3, 2, 2])) and int((~torch.isfinite(y)).sum()) == 1
    assert _sat(reset=True) > 0 and _sat() == 0


@pytest.mark.parametrize('s,p,r', [(1, 1, 32), (2, 0, 65)])
def test_conv_family_on_split_operands_vs_fp64(s, p, r):
    from rick_amd.op import conv as cv, split as sp
    N, ci, co = 4, 256, 512
    x = _octaves((N, ci, r, r), -6, 3, 1)
    w = torch.randn(co, ci, 3, 3, device=DEV)
    ro = (r + 2 * p - 3) // s + 1
    gy = _octaves((N, co, ro, ro), -20, -12, 2)
    xs, gs = sp.split_pack(x), sp.split_pack(gy)
    wp, wpT = cv._pack(w, 1.0), cv._pack(w.transpose(0, 1), 1.0)

    def err(a, ref):
        return float((a.double() - ref).abs().max() / ref.abs().max())
    ref = F.conv2d(x.double(), w.double(), stride=s, padding=p)
    y0 = cv._conv_launch(x, wp, co, 3, 3, s, p)
    y1 = cv._conv_launch(None, wp, co, 3, 3, s, p, x_split=xs)
    assert err(y0, ref) < 2e-6 and err(y1, ref) < 2e-6
    ref = F.conv_transpose2d(gy.double(), w.double(), stride=s, padding=p)
    g1 = cv._convT_launch(None, wpT, ci, 3, 3, s, p, (r, r), x_split=gs)
    assert err(g1, ref) < 2e-6
    wd = w.double().requires_grad_(True)
    wref = torch.autograd.grad(F.conv2d(x.double(), wd, stride=s, padding=p), wd, gy.double())[0]
    for kw in (dict(a_split=gs), dict(b_split=xs), dict(a_split=gs, b_split=xs)):
        gw = cv._wgrad_launch(gy, x, 3, 3, s, p, **kw)
        assert err(gw, wref) < 2e-6, kw.keys()


def test_producers_write_the_same_image_as_the_standalone_pass():
    """FIR / activation adjoint / merge with a split-image result == split_pack(fp32 result) bit for bit under the same bound,
    running maxima are exact, `accumulate` adds."""
    from rick_amd.models import make_kernel
    from rick_amd.op import dblock, split as sp
    from rick_amd.op.fused_act import _ActAdjoint
    from rick_amd.op.upfirdn2d import _fir, _flipped
    _sat(reset=True)
    taps = make_kernel([1, 3, 3, 1]).to(DEV)
    x = _octaves((2, 128, 16, 16), -4, 2, 3)
    A = sp.amax(x)
    # blur (pad 2,2) -> image only
    ref = _fir(x, taps, (1, 1), (1, 1), (2, 2, 2, 2))
    _, img = dblock._fir_ex(x, taps, 1, 1, (2, 2, 2, 2), split_bound=A, no_f32=True)
    assert torch.equal(img.data, sp.split_pack(ref, A).data) and torch.equal(img.hdr[:3], sp.split_pack(ref, A).hdr[:3])
    # decimating FIR (pad 1,1, down 2)
    ref = _fir(x, taps, (1, 1), (2, 2), (1, 1, 1, 1))
    _, img = dblock._fir_ex(x, taps, 1, 2, (1, 1, 1, 1), split_bound=A, no_f32=True)
    assert torch.equal(img.data, sp.split_pack(ref, A).data)
    # zero-insertion adjoint, accumulated into an existing tensor, maximum measured after the addition
    g = _octaves((2, 128, 8, 8), -10, -5, 4)
    base = _octaves((2, 128, 16, 16), -10, -5, 5)
    ref = base + _fir(g, _flipped(taps), (2, 2), (1, 1), (2, 1, 2, 1))
    word = sp.new_amax(DEV)
    out, _ = dblock._fir_ex(g, _flipped(taps), 2, 1, (2, 1, 2, 1), out=base.clone(), amax=word, accumulate=True)
    assert torch.equal(out, ref) and float(sp.amax_value(word)) == float(ref.abs().max())
    # activation adjoint -> two images + bias gradient
    y = _octaves((2, 128, 16, 16), -2, 2, 6)
    gg = _octaves((2, 128, 16, 16), -12, -6, 7)
    Ag = sp.amax(gg)
    c = 1 / math.sqrt(2)
    gz_ref, gb_ref, _ = _ActAdjoint.apply(gg, y, None, 0.2, math.sqrt(2) * c, True, False)
    i1, i2, gb = dblock._act_adjoint_split(gg, y, 0.2, math.sqrt(2) * c, Ag, c, True, None)
    assert torch.equal(i1.data, sp.split_pack(gz_ref, Ag, None, math.sqrt(2) * c).data)
    assert torch.equal(i2.data, sp.split_pack(gg * c, Ag, None, c).data)
    assert torch.equal(gb, gb_ref)
    assert _sat() == 0


def _block(C, O):
    from rick_amd.models import ResBlock
    torch.manual_seed(0)
    blk = ResBlock(C, O).to(DEV)
    with torch.no_grad():
        blk.conv1[1].bias.normal_(0, 0.1)
        blk.conv2[2].bias.normal_(0, 0.1)
    return blk


@pytest.mark.parametrize('N,C,O,H', [(8, 256, 256, 64), (4, 128, 256, 128)])
def test_discriminator_resblock_one_node_vs_per_layer(N, C, O, H):
    """Forward values, input gradient and every parameter gradient of the split-image block against the per-layer path
    (the round-3 path, itself pinned to the reference's goldens) and against an fp64 torch evaluation."""
    from rick_amd import models, op
    from rick_amd.op import dblock, split as sp
    blk = _block(C, O)
    x0 = _octaves((N, C, H, H), -3, 2, 8)
    assert dblock.block_supported(x0, blk.conv1[0].weight, blk.conv2[1].weight, blk.skip[1].weight)
    gout = _octaves((N, O, H // 2, H // 2), -14, -9, 9)
    res = {}
    _sat(reset=True)
    for mode in (True, False):
        models._USE_DBLOCK = mode
        try:
            x = x0.clone().requires_grad_(True)
            feat = []
            out = blk(x, feat)
            params = [blk.conv1[0].weight, blk.conv1[1].bias, blk.conv2[1].weight, blk.conv2[2].bias, blk.skip[1].weight]
            grads = torch.autograd.grad(out, [x] + params, gout)
            res[mode] = [out.detach(), feat[0].detach(), feat[1].detach()] + [g.detach() for g in grads]
        finally:
            models._USE_DBLOCK = True
        if mode:
            assert sp.taken(out, '_rick_split') is not None
    assert _sat() == 0

    def rel(a, b):
        return float((a.detach() - b.detach()).abs().max() / b.detach().abs().max())
    names = ['out', 't1', 't2', 'gx', 'gw1', 'gb1', 'gw2', 'gb2', 'gws']
    for n, a, b in zip(names, res[True], res[False]):
        assert rel(a, b) < 2e-5, (n, rel(a, b))
    # fp64 evaluation of the same block (plain torch): both paths sit at the fp32 noise floor
    k = blk.conv2[0].kernel.double()
    xd = x0.double().requires_grad_(True)
    pd = [p.detach().double().requires_grad_(True) for p in [blk.conv1[0].weight, blk.conv1[1].bias, blk.conv2[1].weight,
                                                              blk.conv2[2].bias, blk.skip[1].weight]]
    sq = math.sqrt(2)

    def blur(t, pad, down=1):
        c = t.shape[1]
        kk = torch.flip(k, [0, 1]).view(1, 1, 4, 4).repeat(c, 1, 1, 1)
        return F.conv2d(F.pad(t, (pad, pad, pad, pad)), kk, groups=c, stride=down)
    t1 = F.leaky_relu(F.conv2d(xd, pd[0] * blk.conv1[0].scale, padding=1) + pd[1].view(1, -1, 1, 1), 0.2) * sq
    t2 = F.leaky_relu(F.conv2d(blur(t1, 2), pd[2] * blk.conv2[1].scale, stride=2) + pd[3].view(1, -1, 1, 1), 0.2) * sq
    sk = F.conv2d(blur(xd, 1, 2), pd[4] * blk.skip[1].scale)
    o = (t2 + sk) / sq
    gr = torch.autograd.grad(o, [xd] + pd, gout.double())
    ref = [o, t1, t2] + list(gr)
    # (L2: an fp32 and an fp64 run may disagree on the sign of a ~0 pre-activation, which moves single gradient entries)
    # — so the yardstick is the per-layer path's own distance from fp64
    for n, a, a0, b in zip(names, res[True], res[False], ref):
        e = float((a.double() - b.detach()).norm() / b.detach().norm())
        e0 = float((a0.double() - b.detach()).norm() / b.detach().norm())
        assert e < 1.5 * e0 + 1e-5 and e < 2e-3, (n, e, e0)


def test_discriminator_chain_passes_images_and_maxima():
    """Two blocks in a row: the second reads the first one's output image, the first one's backward takes the gradient
    maximum from the second (no stand-alone passes in between), values equal the per-layer path's."""
    from rick_amd import models
    from rick_amd.op import split as sp
    b1, b2 = _block(128, 256), _block(256, 512)
    x0 = _octaves((4, 128, 256, 256), -3, 2, 10)
    calls = {'pack': 0, 'amax': 0}
    orig_pack, orig_amax = sp.split_pack, sp.amax

    def count_pack(*a, **k):
        calls['pack'] += 1
        return orig_pack(*a, **k)

    def count_amax(*a, **k):
        calls['amax'] += 1
        return orig_amax(*a, **k)
    res = {}
    for mode in (True, False):
        models._USE_DBLOCK = mode
        sp.split_pack, sp.amax = count_pack, count_amax
        try:
            x = x0.clone().requires_grad_(True)
            y = b2(b1(x))
            loss = (y * y).mean()
            g = torch.autograd.grad(loss, [x, b1.conv1[0].weight, b2.skip[1].weight])
            res[mode] = [y.detach()] + [t.detach() for t in g]
        finally:
            models._USE_DBLOCK = True
            sp.split_pack, sp.amax = orig_pack, orig_amax
        if mode:
            # one stand-alone pack (+ its maximum) for the chain's input, one maximum for the gradient entering the chain
            assert calls == {'pack': 1, 'amax': 2}, calls
    # (two blocks deep a ~0 pre-activation may take the other LeakyReLU branch in the two fp32 evaluations: L2, and the share of
    # entries that moved)
    for i, (a, b) in enumerate(zip(res[True], res[False])):
        e = float((a - b).norm() / b.norm())
        moved = float(((a - b).abs() > 1e-4 * b.abs().max()).float().mean())
        assert e < 1e-3 and moved < 2e-3, (i, e, moved)


@pytest.mark.saturates
def test_on_the_fly_split_counts_what_its_sampled_exponent_misses():
    """The kernels that split fp32 operands themselves take the exponent from a sample of the block's data (MODE.FP16_OVFL on:
    a value ~8 000 x above every sample clamps to 65504, finite and wrong).  The clamp is counted through the hardware's
    tracked operand maximum: a spike 10^5 x above the rest at a position the sample does not visit makes
    rick_saturation_count() non-zero (the forward / data-gradient kernel tracks in the shipping library; the weight-gradient
    and transposed kernels, which have no register to spare, only in the experiment build: csrc/wgrad.hip); the same launches on
    well-scaled data leave it at zero (and so does every other GPU test: tests/conftest.py).  (The counter is fed by two
    v_max3 per four converted elements; the hardware's sticky overflow status would have been free but stays 0 with traps
    disabled on this chip — tools/micro/trapsts.hip.)"""
    from rick_amd.op import conv as cv
    torch.manual_seed(0)
    N, C, H = 4, 256, 64
    w = torch.randn(C, C, 3, 3, device=DEV)
    wp, wpT = cv._pack(w, 1.0), cv._pack(w.transpose(0, 1), 1.0)
    x = _cl(torch.randn(N, C, H, H, device=DEV))
    g = _cl(torch.randn(N, C, H, H, device=DEV))
    assert _sat(reset=True) == 0
    cv._conv_launch(x, wp, C, 3, 3, 1, 1)
    cv._convT_launch(g, wpT, C, 3, 3, 1, 1, (H, H))
    cv._wgrad_launch(g, x, 3, 3, 1, 1)
    torch.cuda.synchronize()
    assert _sat() == 0
    hits = 0
    for trial in range(8):                       # the sampled positions are fixed per block; one of a few spike positions is missed
        xs = x.clone()
        xs[trial % N, 100 + 7 * trial, 17 + trial, 29 - trial] = 1e5
        _sat(reset=True)
        y = cv._conv_launch(xs, wp, C, 3, 3, 1, 1)
        gw = cv._wgrad_launch(g, xs, 3, 3, 1, 1)
        torch.cuda.synchronize()
        n = _sat()
        if n:
            hits += 1
            assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(gw).all())      # clamped, not inf: hence the counter
    assert hits > 0
    _sat(reset=True)


def test_discriminator_input_layer_one_launch_equals_two_op_path():
    """thin product + FusedLeakyReLU of the discriminator's first layer in one launch: values bit-identical to the two-op path,
    the split image it emits equals the stand-alone pass's under the same bound, gradients equal."""
    from rick_amd import models
    from rick_amd.op import split as sp
    torch.manual_seed(2)
    layer = models.ConvLayer(3, 128, 1).to(DEV)
    with torch.no_grad():
        layer[1].bias.normal_(0, 0.2)
    img0 = torch.randn(4, 3, 64, 64, device=DEV)
    res = {}
    for mode in (True, False):
        models._USE_DBLOCK = mode
        try:
            img = img0.clone().requires_grad_(True)
            y = layer(img)
            g = torch.autograd.grad((y * y).sum(), [img, layer[0].weight, layer[1].bias])
            res[mode] = [y.detach()] + [t.detach() for t in g]
            if mode:
                pk = sp.taken(y, '_rick_split')
                assert float(pk.hdr[2]) >= float(y.abs().max())
                assert torch.equal(pk.data, sp.split_pack(y.detach(), *pk.bound).data)
        finally:
            models._USE_DBLOCK = True
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1:], res[False][1:]):
        assert float((a - b).abs().max() / b.abs().max()) < 1e-6


def test_generator_split_image_path_equals_on_the_fly_path():
    """The generator's fused layers hand split images from producer to consumer (op/modconv.py: the conv epilogue / the blur
    write the next layer's operand with its style folded in; the activation adjoint / the blur's adjoint write the backward
    operands with the demodulation folded in).  Image and every parameter gradient against the same network with the images
    switched off (the round-3 path, pinned to the reference's goldens)."""
    from rick_amd.models import Generator
    from rick_amd.op import modconv
    from rick_amd.synth import synth_state_dict
    from tests.shapes import generator_shapes
    size, B = 128, 4
    g = Generator(size, 512, 8)
    g.load_state_dict(synth_state_dict(generator_shapes(size)), strict=False)
    g = g.to(DEV)
    for p_ in g.style.parameters():          # as in the train steps (the optimiser owns no mapping parameter): the modulation bank
        p_.requires_grad_(False)             # evaluates every style up front, which is what lets a layer know the NEXT layer's style
    torch.manual_seed(5)
    z = torch.randn(B, 512, device=DEV)
    noise = [torch.randn(B, 1, n.shape[2], n.shape[3], device=DEV) for n in g.make_noise()]
    params = [p for n, p in g.named_parameters() if n.startswith('convs.') or n.startswith('conv1.')]
    res, used = {}, {}
    for mode in (True, False):
        modconv._USE_SPLIT = mode
        for k in modconv.stats:
            modconv.stats[k] = 0
        try:
            img, _ = g([z], noise=noise)
            loss = (img * torch.linspace(-1, 1, img.numel(), device=DEV).view_as(img)).sum()
            gr = torch.autograd.grad(loss, params, allow_unused=True)
            res[mode] = [img.detach()] + [t.detach() if t is not None else None for t in gr]
            used[mode] = dict(modconv.stats)
        finally:
            modconv._USE_SPLIT = False       # (the shipping default: see op/modconv.py)
    # 128 px: 11 modulated 3x3 layers; the 32^2 ... 128^2 ones (6 layers) run forward, data and weight gradient on images
    assert used[False] == {'fprop': 0, 'dgrad': 0, 'wgrad': 0, 'produced': 0}
    assert used[True]['fprop'] >= 5 and used[True]['dgrad'] >= 5 and used[True]['wgrad'] >= 5, used[True]
    e = float((res[True][0] - res[False][0]).abs().max() / res[False][0].abs().max())
    assert e < 2e-5, e
    for (n, _), a, b in zip([(n, p) for n, p in g.named_parameters() if n.startswith('convs.') or n.startswith('conv1.')], res[True][1:], res[False][1:]):
        assert (a is None) == (b is None), n
        if a is not None:
            e = float((a - b).norm() / (b.norm() + 1e-30))
            assert e < (3e-2 if n.endswith("noise.weight") else 2e-4), (n, e)     # (noise strengths: the reference itself spreads 1e-2)


def test_fir_adjoint_fused_equals_two_passes():
    """Blur adjoint + activation adjoint in one launch (rick_upfirdn2d_ex_f32 with adj_ref): the image equals the two-pass
    result's bit for bit under the same bound (same fmaf chain, same adjoint expression); the bias gradient agrees to fp32
    summation order."""
    from rick_amd.models import make_kernel
    from rick_amd.op import dblock, split as sp
    from rick_amd.op.upfirdn2d import _flipped
    taps = make_kernel([1, 3, 3, 1]).to(DEV)
    flip = _flipped(taps)
    for n, c, h in [(2, 128, 64), (8, 128, 128), (1, 256, 16)]:
        g = _octaves((n, c, h + 1, h + 1), -12, -6, 21)
        y = _octaves((n, c, h, h), -2, 2, 22)
        A = sp.amax(g)
        ref_f, _ = dblock._fir_ex(g, flip, 1, 1, (1, 1, 1, 1))
        ref_img, _, ref_gb = dblock._act_adjoint_split(ref_f, y, 0.2, math.sqrt(2), A, None, True, None)
        img, gb = dblock._fir_adjoint_split(g, flip, (1, 1, 1, 1), y, 0.2, math.sqrt(2), A, True, None)
        assert torch.equal(img.data, ref_img.data) and torch.equal(img.hdr[:3], ref_img.hdr[:3])
        assert float((gb - ref_gb).abs().max() / ref_gb.abs().max()) < 1e-5
        acc = torch.ones(c, device=DEV)
        _, none = dblock._fir_adjoint_split(g, flip, (1, 1, 1, 1), y, 0.2, math.sqrt(2), A, True, acc)
        assert none is None and float((acc - 1 - ref_gb).abs().max() / ref_gb.abs().max()) < 1e-5


def test_hand_over_attributes_go_stale_after_an_in_place_write():
    """`_rick_split` / `_rick_amax` / `_rick_bound` describe a tensor AS IT WAS when the producer attached them
    (op/split.py hand / taken record the version counter and the address).  An in-place write in between — x.mul_(), autograd's
    in-place accumulation, a hook — makes the consumer ignore the attribute and measure / pack again, instead of reading a stale
    image or a maximum that is too small; a view_as alias of the unchanged tensor keeps it (op.torgb_fork)."""
    from rick_amd.op import split as sp
    x = torch.randn(2, 64, 8, 8, device=DEV).contiguous(memory_format=torch.channels_last)
    img = sp.split_pack(x)
    sp.hand(x, '_rick_split', img)
    sp.hand(x, '_rick_amax', img.bound[0])
    assert sp.taken(x, '_rick_split') is img
    alias = x.view_as(x)
    sp.rehand(x, alias)
    assert sp.taken(alias, '_rick_split') is img and sp.taken(alias, '_rick_amax') is img.bound[0]
    x.mul_(4.0)                                   # (alias shares the version counter)
    assert sp.taken(x, '_rick_split') is None and sp.taken(x, '_rick_amax') is None and sp.taken(alias, '_rick_split') is None
    # the consumer side: a discriminator block fed such a tensor packs it again — same result as for a tensor without attributes
    from rick_amd import models
    torch.manual_seed(4)
    blk = models.ResBlock(64, 64).to(DEV)
    x2 = torch.randn(2, 64, 32, 32, device=DEV).contiguous(memory_format=torch.channels_last)
    ref = blk(x2.clone())
    stale = sp.split_pack(x2 * 0.25)              # an image of OTHER values ...
    x3 = x2.clone()
    sp.hand(x3, '_rick_split', stale)
    x3.add_(0.0)                                  # ... invalidated by an in-place write
    assert torch.equal(blk(x3), ref)
