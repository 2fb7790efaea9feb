"""Step-level batching of small launches (round 4): DemodBank (rick_amd/op/modconv.py: the demodulation coefficients of every
generator layer from one launch, wsq cached per weight update), op.deferred_sums (one second-stage launch per backward pass),
op.torgb_fork (the branch-point gradient added by the kernel that produces it), the trainer's latent pool — each one
bit-identical to the path it replaces wherever values are comparable."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gen(size=64):
    from rick_amd.models import Generator
    torch.manual_seed(3)
    g = Generator(size, 512, 2).cuda()
    with torch.no_grad():
        for p in g.parameters():            # spread the weights so that demodulation matters
            if p.ndim == 5:
                p.mul_(torch.rand(p.shape[1], device=p.device).view(1, -1, 1, 1, 1) * 2 + 0.2)
    return g


def _run(g, latent, noise, use_bank, sink=False):
    import rick_amd.models as M
    from rick_amd import op
    M._USE_DEMOD_BANK = use_bank
    try:
        for p in g.parameters():
            p.grad = torch.zeros_like(p) if sink else None
        if sink:
            with op.grad_sink():
                img, _ = g([latent], input_is_latent=True, noise=noise)
                (img * torch.linspace(-1, 1, img.numel(), device=img.device).view_as(img)).sum().backward()
        else:
            img, _ = g([latent], input_is_latent=True, noise=noise)
            (img * torch.linspace(-1, 1, img.numel(), device=img.device).view_as(img)).sum().backward()
        return img.detach().clone(), {n: p.grad.clone() for n, p in g.named_parameters() if p.grad is not None}
    finally:
        M._USE_DEMOD_BANK = True


@pytest.mark.parametrize('sink', [False, True])
def test_demod_bank_equals_per_layer_path_bitwise(sink):
    g = _gen()
    for p in g.style.parameters():
        p.requires_grad_(False)
    B = 4
    latent = torch.randn(B, g.n_latent, 512, device='cuda')
    noise = [torch.randn(1, 1, n.shape[-1], n.shape[-1], device='cuda') for n in g.make_noise()]
    img0, gr0 = _run(g, latent, noise, False, sink)
    img1, gr1 = _run(g, latent, noise, True, sink)
    assert g.__dict__.get('_dmbank') is not None and g._dmbank._stamps is not None      # the bank did run
    assert torch.equal(img0, img1)
    assert gr0.keys() == gr1.keys() and len(gr0) > 20
    for k in gr0:
        assert torch.equal(gr0[k], gr1[k]), k


def test_demod_bank_coefficients_and_wsq_refresh():
    """d of every layer == demod_coeff_fused (bitwise) and == the tensor-algebra form (1e-6); after an in-place weight update the
    cached wsq is recomputed (stamp = parameter version / weights epoch), not reused."""
    from rick_amd.op import modconv as mc
    g = _gen()
    B = 3
    latent = torch.randn(B, g.n_latent, 512, device='cuda')
    with torch.no_grad():
        sb = g._modulation_bank()(latent)
        bank = g._demod_bank()
        for rnd in range(2):
            dl = bank(sb)
            assert dl is not None and len(dl) == len(bank.convs)
            for c, j, d in zip(bank.convs, bank.s_index, dl):
                w = c.weight.view(c.out_channel, c.in_channel, c.kernel_size, c.kernel_size)
                ref = mc.demod_coeff_fused(w, sb[j], c.scale, c.eps)
                assert torch.equal(d, ref)
                torch.testing.assert_close(d, mc.demod_coeff(w.double(), sb[j].double(), c.scale, c.eps).float(), rtol=2e-6, atol=0)
            launched = bank._stamps
            bank(sb)
            assert bank._stamps is launched or bank._stamps == launched          # nothing changed: wsq not recomputed
            for c in bank.convs[::3]:
                c.weight.mul_(1.25)                                               # in place: bumps the version counter


def test_demod_bank_create_graph_falls_back_to_tensor_algebra():
    """A create_graph=True backward through the bank differentiates the per-layer tensor algebra: the second derivative of the
    image w.r.t. a convolution weight through d exists and matches the per-layer path."""
    import rick_amd.models as M
    g = _gen(32)
    for p in g.style.parameters():
        p.requires_grad_(False)
    latent = torch.randn(2, g.n_latent, 512, device='cuda')
    noise = [torch.randn(1, 1, n.shape[-1], n.shape[-1], device='cuda') for n in g.make_noise()]
    w = g.convs[0].conv.weight
    out = []
    for use in (False, True):
        M._USE_DEMOD_BANK = use
        try:
            img, _ = g([latent], input_is_latent=True, noise=noise)
            (g1,) = torch.autograd.grad(img.square().mean(), w, create_graph=True)
            (g2,) = torch.autograd.grad(g1.square().sum(), w)
            out.append((g1.detach(), g2))
        finally:
            M._USE_DEMOD_BANK = True
    torch.testing.assert_close(out[0][0], out[1][0], rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(out[0][1], out[1][1], rtol=2e-3, atol=1e-9)


def test_trainer_graph_steps_bank_equals_per_layer_bitwise():
    """Four graph-replayed iterations (D, R1, G, path length, EMA) with the demodulation bank leave exactly the generator,
    discriminator, EMA weights, Adam moments and losses the per-layer kernels leave: the cached wsq is refreshed behind every weight
    update (PackGroup.after_repack) and the banked backward adds the same gradients."""
    import rick_amd.models as M
    from rick_amd.synth import synth_reals, synth_tensor
    from rick_amd.train import RickTrainer, TrainConfig
    from tests.test_gpu_models import build
    size, B = 32, 2
    real = [synth_reals(B, size=size, seed=170 + k).cuda() for k in range(4)]
    lat = {k: synth_tensor(f'dmbank/lat/{k}', (B if k != 'plr' else 1, 8, 512)).cuda() for k in ('d', 'g', 'plr')}
    lat['plr'].requires_grad_(True)
    pl_noise = synth_tensor('dmbank/pl', (1, 3, size, size)).cuda()

    def run(use_bank):
        M._USE_DEMOD_BANK = use_bank
        try:
            g, d = build(size)
            tr = RickTrainer(TrainConfig(size=size, batch=B, warmup_iter=0), g, d, *build(size))
            noises = [getattr(g.noises, f'noise_{i}') for i in range(g.num_layers)]
            tr.enable_graphs(True)
            tr._draw_inject('d')
            tr._graph_latents = lambda key, batch: lat[key]
            static_real = torch.empty_like(real[0])
            for k in range(4):
                static_real.copy_(real[k])
                tr.d_step(static_real, None, g_noise=noises, graph=True)
                tr.r1_step(static_real, graph=True)
                tr.g_step(None, g_noise=noises, graph=True)
                tr.plr_step(None, pl_noise=pl_noise, g_noise=noises, graph=True)
                tr.ema_step()
            torch.cuda.synchronize()
            assert ('_dmbank' in g.__dict__) == use_bank
            return tr
        finally:
            M._USE_DEMOD_BANK = True
    a, b = run(False), run(True)
    for fa, fb in ((a.g_flat, b.g_flat), (a.d_flat, b.d_flat), (a.g_ema_flat, b.g_ema_flat)):
        assert torch.equal(fa.flat, fb.flat)
    assert torch.equal(a.g_optim.m, b.g_optim.m) and torch.equal(a.g_optim.v, b.g_optim.v)
    for k in ('d', 'g', 'r1', 'path'):
        assert torch.equal(a.losses[k], b.losses[k])


def test_trainer_graph_steps_deferred_sums_equal_immediate_bitwise():
    """op.deferred_sums(): the bias / noise-strength column sums of a whole backward pass from one launch at its end leave
    exactly the weights, moments and losses the per-layer second stages leave (four graph-replayed iterations)."""
    import rick_amd.op.fused_act as fa
    from rick_amd.synth import synth_reals, synth_tensor
    from rick_amd.train import RickTrainer, TrainConfig
    from tests.test_gpu_models import build
    size, B = 32, 2
    real = [synth_reals(B, size=size, seed=270 + k).cuda() for k in range(4)]
    lat = {k: synth_tensor(f'defer/lat/{k}', (B if k != 'plr' else 1, 8, 512)).cuda() for k in ('d', 'g', 'plr')}
    lat['plr'].requires_grad_(True)
    pl_noise = synth_tensor('defer/pl', (1, 3, size, size)).cuda()
    launched = []
    orig = fa.flush_colsums

    def run(off):
        fa._DEFER_OFF = off
        fa.flush_colsums = lambda items: (launched.append(len(items or [])), orig(items))[1]
        try:
            g, d = build(size)
            tr = RickTrainer(TrainConfig(size=size, batch=B, warmup_iter=0), g, d, *build(size))
            noises = [getattr(g.noises, f'noise_{i}') for i in range(g.num_layers)]
            tr.enable_graphs(True)
            tr._draw_inject('d')
            tr._graph_latents = lambda key, batch: lat[key]
            static_real = torch.empty_like(real[0])
            for k in range(4):
                static_real.copy_(real[k])
                tr.d_step(static_real, None, g_noise=noises, graph=True)
                tr.r1_step(static_real, graph=True)
                tr.g_step(None, g_noise=noises, graph=True)
                tr.plr_step(None, pl_noise=pl_noise, g_noise=noises, graph=True)
                tr.ema_step()
            torch.cuda.synchronize()
            return tr
        finally:
            fa._DEFER_OFF = False
            fa.flush_colsums = orig
    a = run(True)
    assert max(launched, default=0) == 0
    b = run(False)
    assert max(launched) >= 5                  # several layers' sums per flush
    for fa_, fb_ in ((a.g_flat, b.g_flat), (a.d_flat, b.d_flat), (a.g_ema_flat, b.g_ema_flat)):
        assert torch.equal(fa_.flat, fb_.flat)
    assert torch.equal(a.g_optim.m, b.g_optim.m) and torch.equal(a.d_optim.v, b.d_optim.v)
    for k in ('d', 'g', 'r1', 'path'):
        assert torch.equal(a.losses[k], b.losses[k])


@pytest.mark.parametrize('sink', [False, True])
def test_torgb_fork_adds_the_branch_gradient_in_the_kernel(sink):
    """op.torgb_fork: the activation that feeds ToRGB and the next layer is ONE autograd node with two outputs, so ToRGB's data
    gradient is added into the next layer's by rick_torgb_bwdx_acc_f32 — image and every gradient torch.equal to the path on
    which autograd sums the two (verdict round 3, item 4)."""
    import rick_amd.models as M
    g = _gen()
    for p in g.style.parameters():
        p.requires_grad_(False)
    latent = torch.randn(3, g.n_latent, 512, device='cuda')
    noise = [torch.randn(1, 1, n.shape[-1], n.shape[-1], device='cuda') for n in g.make_noise()]
    out = []
    for use in (False, True):
        M._USE_RGB_FORK = use
        try:
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                res = _run(g, latent, noise, True, sink)
                torch.cuda.synchronize()
            adds = sum(ev.count for ev in prof.key_averages() if 'CUDAFunctor_add' in ev.key)
            out.append((res, adds))
        finally:
            M._USE_RGB_FORK = True
    (img0, gr0), adds0 = out[0]
    (img1, gr1), adds1 = out[1]
    assert torch.equal(img0, img1)
    assert gr0.keys() == gr1.keys()
    for k in gr0:
        assert torch.equal(gr0[k], gr1[k]), k
    assert adds1 <= adds0 - 3, (adds0, adds1)      # one big add per resolution below the last is gone (64 px: 4 of them)


def test_torgb_fork_never_modifies_a_gradient_it_does_not_own():
    """the in-place accumulate is taken only into a buffer the producing backward marked as exclusively owned
    (the modulated convolution's fresh data gradient).  A gradient that arrives from anywhere else — here from torch's own
    multiplication backward, which carries no mark — is copied first, and the result equals the unfused form."""
    from rick_amd import op
    from rick_amd.op import misc
    torch.manual_seed(5)
    n, c, h = 2, 64, 16
    x = torch.randn(n, c, h, h, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(3, c, device='cuda', requires_grad=True)
    s = torch.rand(n, c, device='cuda') + 0.5
    const = torch.randn(n, c, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
    keep = const.clone()
    before = dict(misc.stats)
    xo, t = op.torgb_fork(x, w, s, wscale=0.125)
    ((xo * const).sum() + (t * t).sum()).backward()
    gx_fork, gw_fork = x.grad.clone(), w.grad.clone()
    assert torch.equal(const, keep)
    assert misc.stats['fork_copy'] == before['fork_copy'] + 1 and misc.stats['fork_inplace'] == before['fork_inplace']
    x.grad = w.grad = None
    t2 = op.torgb(x, w, s, wscale=0.125)
    ((x * const).sum() + (t2 * t2).sum()).backward()
    assert torch.equal(t, t2) and torch.equal(gw_fork, w.grad)
    assert torch.allclose(gx_fork, x.grad, rtol=0, atol=1e-6 * float(x.grad.abs().max()))   # a + b vs b + a: same fp32 sum
    # the generator's own hand-over is the owned, in-place one
    g = _gen()
    for p in g.style.parameters():
        p.requires_grad_(False)
    latent = torch.randn(2, g.n_latent, 512, device='cuda')
    noise = [torch.randn(1, 1, m.shape[-1], m.shape[-1], device='cuda') for m in g.make_noise()]
    before = dict(misc.stats)
    _run(g, latent, noise, True, False)
    assert misc.stats['fork_inplace'] >= before['fork_inplace'] + 3 and misc.stats['fork_copy'] == before['fork_copy']


def test_latent_pool_serves_fresh_rows_and_refills():
    """RickTrainer's latent pool: with a frozen mapping network the W-space rows of the next LATENT_POOL steps come from one
    pass through the mapping layers; a graph-replayed step gathers its own rows by a device-side index.  Every step sees different
    latents, the pool is refilled when used up, rows are exactly style(z) of the pool's noise, and a trainable mapping network
    switches the pool off."""
    from rick_amd.synth import synth_reals
    from rick_amd.train import RickTrainer, TrainConfig
    from tests.test_gpu_models import build
    size, B = 32, 2
    g, d = build(size)
    tr = RickTrainer(TrainConfig(size=size, batch=B, warmup_iter=0), g, d, *build(size))
    assert tr._pool_ok()
    tr.enable_graphs(True)
    real = synth_reals(B, size=size, seed=5).cuda()
    seen = []
    orig = g.forward

    def spy(styles, **kw):
        if kw.get('input_is_latent'):
            seen.append(styles[0].detach().clone())
        return orig(styles, **kw)
    g.forward = spy                      # (eager warm-up steps only: a replayed graph does not call Python)
    P = tr.LATENT_POOL
    pool_rows = []
    for k in range(P + 4):
        tr.d_step(real, None, graph=True)
        ent = tr._lat_pool['d']
        pool_rows.append((int(ent['idx']), ent['w'][int(ent['idx'])].clone()))
    torch.cuda.synchronize()
    g.forward = orig
    assert [i for i, _ in pool_rows] == [k % P for k in range(P + 4)]                  # advances, wraps after a refill
    rows = torch.stack([r for _, r in pool_rows])
    assert len({tuple(r.flatten()[:8].tolist()) for r in rows}) == P + 4               # all different (refilled, not re-used)
    assert torch.isfinite(rows).all() and 0.1 < float(rows.std()) < 10
    # what the eager steps fed the generator is the mixed view of exactly those rows
    assert len(seen) >= 2
    for lat, (_, r) in zip(seen[:2], pool_rows):          # (the third call is the capture: its clone only has values at replay)
        w = r.view(2, B, -1)
        assert all(bool((lat[:, j] == w[0]).all() or (lat[:, j] == w[1]).all()) for j in range(lat.shape[1]))
    for p_ in g.style.parameters():
        p_.requires_grad_(True)
    assert not tr._pool_ok()


def test_latent_pool_follows_reloaded_mapping_network():
    """pooled W rows are values of the mapping network AS IT WAS when the pool was filled.  Loading other
    mapping weights into a trainer that has already stepped (checkpoint.resume / load_source) must not leave up to 15 steps
    training on rows of the old network: the next step refills the pool from the loaded weights (in place — a captured step
    keeps gathering from the same buffer), and invalidate_graphs() / resume() drop the pool with the graphs."""
    from rick_amd import checkpoint
    from rick_amd.synth import synth_reals
    from rick_amd.train import RickTrainer, TrainConfig
    from tests.test_gpu_models import build
    size, B = 32, 2
    g, d = build(size)
    tr = RickTrainer(TrainConfig(size=size, batch=B, warmup_iter=0), g, d, *build(size))
    tr.enable_graphs(True)
    real = synth_reals(B, size=size, seed=5).cuda()
    for _ in range(4):                                   # two eager steps, the capture, one replay
        tr.d_step(real, None, graph=True)
    ent = tr._lat_pool['d']
    assert int(ent['idx']) == 3
    buf = ent['w'].data_ptr()
    sd = {k: v.clone() for k, v in g.state_dict().items()}
    for k in sd:
        if k.startswith('style.') and k.endswith('weight'):
            sd[k] = sd[k] * 1.25
    g.load_state_dict(sd)
    torch.manual_seed(11)
    tr.d_step(real, None, graph=True)                    # a replay; its host side refilled the pool first
    torch.cuda.synchronize()
    assert tr._lat_pool['d'] is ent and ent['w'].data_ptr() == buf and int(ent['idx']) == 0
    torch.manual_seed(11)
    with torch.no_grad():
        z = torch.randn(tr.LATENT_POOL * 2 * B, 512, device='cuda')
        want = g.style(z).view(tr.LATENT_POOL, 2 * B, -1)
    assert torch.equal(ent['w'], want)
    # resume(): graphs and pool go together; the next eager step builds a fresh pool from the restored weights
    ck = checkpoint.state_dict(tr)
    checkpoint.resume(tr, ck)
    assert '_lat_pool' not in tr.__dict__
    torch.manual_seed(12)
    tr.d_step(real, None, graph=True)
    torch.manual_seed(12)
    with torch.no_grad():
        z = torch.randn(tr.LATENT_POOL * 2 * B, 512, device='cuda')
        want = g.style(z).view(tr.LATENT_POOL, 2 * B, -1)
    assert torch.equal(tr._lat_pool['d']['w'], want)


def test_back_to_back_captures_do_not_share_running_maxima():
    """running maxima / image headers come from an arena of zero-filled words; a graph must contain the fill
    of the words ITS launches accumulate into.  Steps captured back to back (prepare_graphs, the
    loop below) used to share one arena block, zero-filled by the first graph only, so a G replay that did not follow a D replay kept
    the previous replay's maxima (looser exponents: results no longer equal to eager).  Here: two G replays in a row, the first
    on 4x larger latents; the second must equal an eager G step on its own latents bit for bit."""
    from rick_amd.op import split as sp
    from rick_amd.synth import synth_reals, synth_tensor
    from rick_amd.train import RickTrainer, TrainConfig
    from tests.test_gpu_models import build
    size, B = 32, 2
    real = synth_reals(B, size=size, seed=5).cuda()

    def trainer():
        g, d = build(size)
        tr = RickTrainer(TrainConfig(size=size, batch=B, warmup_iter=0), g, d, *build(size))
        tr.g_optim.lr = tr.d_optim.lr = 0.0            # parameters stay put: repeated steps are comparable
        tr.enable_graphs(True)
        lat = torch.zeros(B, g.n_latent, 512, device='cuda')
        tr._graph_latents = lambda key, batch: lat[:batch]
        return tr, lat
    small = synth_tensor('arena/lat', (B, 8, 512)).cuda()              # n_latent = 8 at 32 px
    a, lat_a = trainer()
    noises = [synth_tensor(f'arena/noise/{i}', tuple(getattr(a.g.noises, f'noise_{i}').shape)).cuda() for i in range(a.g.num_layers)]
    lat_a.copy_(small)
    for _ in range(3):                                  # third round: the D capture and the G capture are back to back
        a.d_step(real, None, g_noise=noises, graph=True)
        a.g_step(None, g_noise=noises, graph=True)
    assert all('graphs' in a._gs[k] for k in ('d', 'g'))
    lat_a.copy_(small * 4)
    a.g_step(None, g_noise=noises, graph=True)          # replay on large latents
    lat_a.copy_(small)
    a.g_step(None, g_noise=noises, graph=True)          # replay again, no D replay in between
    got = a.g_flat.grad.clone()
    b, lat_b = trainer()
    lat_b.copy_(small)
    b.g_step(None, g_noise=noises, graph=True)          # first call of a step type: eager
    assert 'graphs' not in b._gs['g']
    assert torch.equal(got, b.g_flat.grad)
    assert sp.capture_id() == 0


@pytest.mark.parametrize('graphs', [False, True], ids=['eager', 'graphs'])
def test_wgrad_overlap_leaves_every_value_unchanged(graphs):
    """op.wgrad_overlap(): the sunk weight-gradient launches of a D / G step run on a second stream next to the data-gradient
    chain (fork / join captured into the step graphs).  Same launches on the same data: parameters, moments and losses after
    D, G, D, G steps are bit-identical to the one-stream run, eagerly and replayed."""
    from rick_amd.op import conv as cv
    from rick_amd.synth import synth_reals
    from rick_amd.train import RickTrainer, TrainConfig
    from tests.test_gpu_models import build
    size, B = 64, 4
    real = synth_reals(B, size=size, seed=5).cuda()
    out = []
    for off in (True, False):
        cv._OVERLAP_OFF = off
        try:
            import random
            random.seed(2)
            torch.manual_seed(2)
            g, d = build(size)
            tr = RickTrainer(TrainConfig(size=size, batch=B, warmup_iter=0), g, d, *build(size))
            tr.enable_graphs(graphs)
            launched = cv._side['pending']
            for k in range(4 if graphs else 2):
                tr.d_step(real, None if graphs else [torch.randn(B, 512, device='cuda')], graph=graphs)
                tr.g_step(None if graphs else [torch.randn(B, 512, device='cuda')], graph=graphs)
            torch.cuda.synchronize()
            assert not cv._side['pending'] and not launched          # every fork was joined
            out.append([t.clone() for t in (tr.g_flat.flat, tr.d_flat.flat, tr.g_optim.m, tr.d_optim.v, tr.losses['d'], tr.losses['g'])])
        finally:
            cv._OVERLAP_OFF = True
    assert cv._side['stream'] is not None                            # the overlapped run did use the side stream
    cv._OVERLAP_OFF = True                                           # (the default: measured slower, see op/conv.py)
    for a, b in zip(*out):
        assert torch.equal(a, b)


def test_deferred_sums_with_a_module_applied_twice_and_from_a_second_thread():
    """the items of one op.deferred_sums() flush run as blocks of ONE launch, each doing a plain read-add-write of its
    destination — a FusedLeakyReLU applied twice in one backward (D called on real and fake separately, a shared bias) queues the same
    .grad twice: the first item is launched before the second is queued, so the result equals the immediate path bit for bit.  And the
    context is one-backward-at-a-time: a second thread that tries to open it while it is open gets a RuntimeError, not another
    thread's sums."""
    import threading
    from rick_amd import op
    torch.manual_seed(8)
    act = op.FusedLeakyReLU(64).cuda()
    with torch.no_grad():
        act.bias.normal_()
    xa = torch.randn(2, 64, 16, 16, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xb = torch.randn(2, 64, 16, 16, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
    res = []
    for deferred in (False, True):
        act.bias.grad = torch.zeros_like(act.bias)
        xa.grad = xb.grad = None
        with op.grad_sink():
            loss = (act(xa) * 1.5).sum() + (act(xb) * -0.5).sum()       # the same bias twice
            if deferred:
                with op.deferred_sums():
                    loss.backward()
            else:
                loss.backward()
        res.append((act.bias.grad.clone(), xa.grad.clone(), xb.grad.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert float(res[0][0].abs().max()) > 0
    err = []

    def other():
        try:
            with op.deferred_sums():
                pass
        except RuntimeError as e:
            err.append(e)
    with op.deferred_sums():
        t = threading.Thread(target=other)
        t.start()
        t.join()
    assert len(err) == 1 and 'another thread' in str(err[0])
    with op.deferred_sums():            # and it is usable again afterwards
        pass
