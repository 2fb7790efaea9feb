"""Row (e) on one GPU: two ranks of the RICK loop share cuda:0 and exchange gradients through
DataParallelGrads (bucketed, launched from post-accumulate hooks; gloo with host staging, because
RCCL refuses two ranks on one device).  The result must be bit-identical to the same two ranks
using a trivial blocking all-reduce of the whole flat buffer, across the warm-up -> full-D stage
change, the frozen-D G step and the second-order R1 / path-length steps."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _get(q, procs, timeout):
    """q.get that fails as soon as a worker has died (an exception in a rank must not cost the whole timeout)."""
    import queue
    import time
    t0 = time.time()
    while True:
        try:
            return q.get(timeout=2)
        except queue.Empty:
            dead = [p.exitcode for p in procs if not p.is_alive() and p.exitcode not in (0, None)]
            if dead:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                raise RuntimeError(f'worker exited with code {dead[0]} before reporting') from None
            if time.time() - t0 > timeout:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                raise


class _BlockingDP:
    """Reference exchange: no hooks, no buckets — average the whole gradient buffer after backward."""

    hooks_enabled = True     # like the bucketed exchange in eager mode: the trainer then routes parameter gradients through
                             # autograd (no gradient sink), so both runs execute the same arithmetic up to the exchange

    def __init__(self):
        self.world = torch.distributed.get_world_size()

    def attach(self, *flats):
        pass

    def prepare(self, flat):
        pass

    def all_reduce(self, flat):
        host = flat.grad.cpu()
        torch.distributed.all_reduce(host)
        flat.grad.copy_(host)
        flat.grad.mul_(1.0 / self.world)


def _worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    from rick_amd.dist import DataParallelGrads, init_from_env
    from rick_amd.models import Discriminator, Generator
    from rick_amd.synth import synth_latents, synth_reals, synth_state_dict, synth_tensor
    from rick_amd.train import RickTrainer, TrainConfig
    from tests.shapes import discriminator_shapes, generator_shapes
    init_from_env('gloo')
    size, B, dev = 32, 2, 'cuda:0'

    def build():
        g = Generator(size, 512, 8, channel_multiplier=2)
        d = Discriminator(size, channel_multiplier=2)
        g.load_state_dict(synth_state_dict(generator_shapes(size)), strict=False)
        d.load_state_dict(synth_state_dict(discriminator_shapes(size)), strict=False)
        return g.to(dev), d.to(dev)
    g, d = build()
    g_ema, d_ema = build()
    dp = DataParallelGrads(bucket_bytes=256 * 1024) if mode == 'bucketed' else _BlockingDP()
    tr = RickTrainer(TrainConfig(size=size, batch=B, warmup_iter=1), g, d, g_ema, d_ema, dp=dp)
    if mode == 'bucketed':
        assert len(dp._state[id(tr.d_flat)]['buckets']) >= 3
    z = synth_latents(B, seed=100 + rank).to(dev)                 # each rank its own micro-batch
    real = synth_reals(B, size=size, seed=200 + rank).to(dev)
    noises = [synth_tensor(f'dpnoise/{i}', tuple(getattr(g.noises, f'noise_{i}').shape)).to(dev)
              for i in range(g.num_layers)]
    pl_noise = synth_tensor(f'dp/pl{rank}', (1, 3, size, size)).to(dev)
    tr.d_step(real, [z], i=0, g_noise=noises)                     # warm-up: only final_* of D carries gradients
    tr.r1_step(real, i=0)
    tr.d_step(real, [z], i=1, g_noise=noises)                     # stage change: every D bucket now fills
    tr.g_step([z], g_noise=noises)                                # D frozen
    tr.plr_step([z[:1]], pl_noise=pl_noise, g_noise=noises)
    tr.d_step(real, [z], i=2, g_noise=noises)
    torch.cuda.synchronize()
    q.put((rank, tr.g_flat.flat.detach().cpu().numpy(), tr.d_flat.flat.detach().cpu().numpy()))
    torch.distributed.destroy_process_group()


def _run(mode):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([_get(q, procs, 600) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return outs


def describe_difference(net, x, y, size=32):
    """Per parameter: how many elements of two flat parameter buffers differ and by how much (names the layer)."""
    from rick_amd.models import Discriminator, Generator
    from rick_amd.train import FlatParams, d_optim_filter, g_optim_filter
    mod, flt = ((Generator(size, 512, 8), g_optim_filter) if net == 1 else (Discriminator(size), d_optim_filter))
    fp = FlatParams(mod.named_parameters(), flt)
    lines = []
    for n in fp.names:
        lo, hi = fp.segment(n)
        ne = int((x[lo:hi] != y[lo:hi]).sum())
        if ne:
            lines.append(f'{n}: {ne} of {hi - lo} differ, max |d| {np.abs(x[lo:hi] - y[lo:hi]).max():.3e} '
                         f'(max |p| {np.abs(y[lo:hi]).max():.3e})')
    return '\n'.join(lines)


def test_two_rank_trainer_bucketed_equals_blocking_allreduce():
    a = _run('bucketed')
    b = _run('blocking')
    for net in (1, 2):
        assert np.array_equal(a[0][net], a[1][net]), describe_difference(net, a[0][net], a[1][net])   # replicas stay identical
        assert np.isfinite(a[0][net]).all()
        assert np.array_equal(a[0][net], b[0][net]), describe_difference(net, a[0][net], b[0][net])   # same update as the plain exchange


# ------------------------------------------------------------------------------------------------------------------
# The data-parallel invariant for the REAL trainer: the gradients two ranks hold after the bucketed
# exchange equal the gradients ONE process computes on the concatenated global batch when minibatch-stddev keeps
# per-rank groups (nn.DataParallel's per-device chunks, train_dynamic_update_prune.py:941-944) — for the D step, the G step
# and (graph mode) the split forward/backward-graph -> all-reduce -> optimiser-graph form; and a Fisher sweep sharded over
# the ranks yields bit-identical masks on every rank, equal to the single-process sweep over all samples.
def _inv_inputs(size, B, world):
    from rick_amd.synth import synth_latents, synth_reals, synth_tensor
    z = [synth_latents(B, seed=300 + r) for r in range(world)]
    real = [synth_reals(B, size=size, seed=400 + r) for r in range(world)]
    fz = [synth_latents(1, seed=500 + j) for j in range(4)]
    fr = [synth_reals(1, size=size, seed=600 + j) for j in range(4)]
    return z, real, fz, fr


def _inv_build(size, dev):
    from rick_amd.models import Discriminator, Generator
    from rick_amd.synth import synth_state_dict
    from tests.shapes import discriminator_shapes, generator_shapes
    g = Generator(size, 512, 8, channel_multiplier=2)
    d = Discriminator(size, channel_multiplier=2)
    g.load_state_dict(synth_state_dict(generator_shapes(size)), strict=False)
    d.load_state_dict(synth_state_dict(discriminator_shapes(size)), strict=False)
    return g.to(dev), d.to(dev)


def _inv_worker(rank, world, port, graphs, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    from rick_amd.dist import DataParallelGrads, init_from_env
    from rick_amd.train import RickTrainer, TrainConfig
    init_from_env('gloo')
    size, B, dev = 32, 2, 'cuda:0'
    g, d = _inv_build(size, dev)
    tr = RickTrainer(TrainConfig(size=size, batch=B, warmup_iter=0, num_fisher_img=4, prune_quantile=1.0), g, d,
                     *_inv_build(size, dev), dp=DataParallelGrads(bucket_bytes=256 * 1024))
    z, real, fz, fr = _inv_inputs(size, B, world)
    noises = [getattr(g.noises, f'noise_{i}') for i in range(g.num_layers)]
    out = {}
    # Fisher sweep: samples j = rank, rank + world, ... on this rank; per-filter vectors summed over ranks
    mine = list(range(rank, 4, world))
    tr.fisher_sweep([fz[j].to(dev) for j in mine], [fr[j].to(dev) for j in mine], first=True, fixed_noise=True)
    out['mask_g'], out['mask_d'] = tr.g_optim.mask.cpu().numpy(), tr.d_optim.mask.cpu().numpy()
    tr.g_optim.mask.zero_()
    tr.d_optim.mask.zero_()
    # keep parameters where they are (the gradients of repeated steps stay comparable): learning rate 0
    tr.d_optim.lr = 0.0
    tr.g_optim.lr = 0.0
    if graphs:
        tr.enable_graphs(True)
        lat = {'d': tr.g.style(z[rank].to(dev)).unsqueeze(1).repeat(1, g.n_latent, 1).detach(),
               'g': tr.g.style(z[rank].to(dev)).unsqueeze(1).repeat(1, g.n_latent, 1).detach()}
        tr._draw_inject('d')
        tr._draw_inject('g')
        tr._graph_latents = lambda key, batch: lat[key]
        static_real = real[rank].to(dev).clone()
        for _ in range(4):                                       # two eager runs, the capture, one replay
            tr.d_step(static_real, None, g_noise=noises, graph=True)
        tr._finish_pending()                                     # the step's exchange + optimiser part are deferred (pipelining)
        out['d_grad'] = tr.d_flat.grad.cpu().numpy()
        for _ in range(4):
            tr.g_step(None, g_noise=noises, graph=True)
        tr._finish_pending()
        out['g_grad'] = tr.g_flat.grad.cpu().numpy()
        gh, g1, g2 = tr._gs['d']['graphs']
        assert gh is None and g1 is not None and g2 is not None          # forward/backward | exchange | optimiser
        gh, g1, g2 = tr._gs['g']['graphs']
        assert gh is not None and g2 is not None                         # generator forward | D forward + backward | exchange | optimiser
    else:
        tr.d_step(real[rank].to(dev), [z[rank].to(dev)], g_noise=noises)
        out['d_grad'] = tr.d_flat.grad.cpu().numpy()
        tr.g_step([z[rank].to(dev)], g_noise=noises)
        out['g_grad'] = tr.g_flat.grad.cpu().numpy()
    torch.cuda.synchronize()
    q.put((rank, out))
    torch.distributed.destroy_process_group()


def _inv_run(graphs):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_inv_worker, args=(r, 2, port, graphs, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([_get(q, procs, 900) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return [o[1] for o in outs]


@pytest.mark.parametrize('graphs', [False, True], ids=['eager_buckets', 'split_graphs'])
def test_two_rank_gradients_equal_single_process_global_batch(graphs):
    from rick_amd.train import RickTrainer, TrainConfig, d_logistic_loss, g_nonsaturating_loss
    a, b = _inv_run(graphs)
    for k in ('d_grad', 'g_grad', 'mask_g', 'mask_d'):
        assert np.array_equal(a[k], b[k]), k                    # identical on both ranks (gradients AND masks)
    # ---- single process, global batch 2 x B, minibatch-stddev in per-rank groups
    size, B, world, dev = 32, 2, 2, 'cuda:0'
    g, d = _inv_build(size, dev)
    tr = RickTrainer(TrainConfig(size=size, batch=B * world, warmup_iter=0, num_fisher_img=4, prune_quantile=1.0), g, d,
                     *_inv_build(size, dev))
    z, real, fz, fr = _inv_inputs(size, B, world)
    noises = [getattr(g.noises, f'noise_{i}') for i in range(g.num_layers)]
    tr.fisher_sweep([t.to(dev) for t in fz], [t.to(dev) for t in fr], first=True, fixed_noise=True)
    for got, ref in ((a['mask_g'], tr.g_optim.mask.cpu().numpy()), (a['mask_d'], tr.d_optim.mask.cpu().numpy())):
        # (per-filter sums are added in a different order across ranks: a filter sitting exactly on a percentile line may move)
        assert (got != ref).mean() < 2e-3
    zc, rc = torch.cat(z).to(dev), torch.cat(real).to(dev)
    with torch.no_grad():
        fake, _ = g([zc], noise=noises)
    tr._set_d_stage(10 ** 9)
    pred, _ = d(torch.cat([fake, rc], 0), calls=2 * world)       # [fake r0 | fake r1 | real r0 | real r1], one stddev group per rank and call
    fake_pred, real_pred = pred.chunk(2, 0)
    tr.d_flat.zero_grad()
    d_logistic_loss(real_pred, fake_pred).backward()
    ref = tr.d_flat.grad.cpu().numpy()
    # (batch 2 per rank vs batch 4 in one pass: other tiles, split-K and summation orders of the bf16x3 kernels)
    assert np.abs(a['d_grad'] - ref).max() <= 5e-4 * np.abs(ref).max(), np.abs(a['d_grad'] - ref).max() / np.abs(ref).max()
    assert np.linalg.norm(a['d_grad'] - ref) <= 1e-3 * np.linalg.norm(ref)
    fake, _ = g([zc], noise=noises)
    with tr._d_frozen():
        fp, _ = d(fake, calls=world)
        tr.g_flat.zero_grad()
        g_nonsaturating_loss(fp).backward()
    ref = tr.g_flat.grad.cpu().numpy()
    # (the G gradient has passed through D and G: a pre-activation within rounding of 0 may take the other LeakyReLU branch
    # in the other tiling — isolated entries move by ~1e-3 of the largest gradient, the L2 criterion below stays tight)
    assert np.abs(a['g_grad'] - ref).max() <= 2e-3 * np.abs(ref).max(), np.abs(a['g_grad'] - ref).max() / np.abs(ref).max()
    assert np.linalg.norm(a['g_grad'] - ref) <= 1e-3 * np.linalg.norm(ref)


def test_bench_self_launch_two_ranks_one_gpu():
    """`python bench.py --gpus 2` with no launcher: the script starts two rank processes itself (here both on cuda:0 with
    gloo + host staging, because RCCL refuses two ranks on one device), runs the real benchmark loop at a small size
    through the split-graph data-parallel path and prints one JSON line; exit code 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RICK_DIST_BACKEND='gloo', RICK_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--size', '32', '--batch', '2',
                        '--steps', '4', '--warmup', '2', '--no-cpu-baseline', '--fisher-img', '2'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['value'] > 0 and out['config']['global_batch'] == 4
    assert out['ranks']['world_size'] == 2 and len(out['ranks']['ms_per_step_per_rank']) == 2
    assert out['ranks']['backend'] == 'gloo' and out['ranks']['rccl_ranks'] == 0


def _nccl_worker(q, direct_steps=False, size=32, B=2, bucket_bytes=256 * 1024, iters=9):
    """Single rank, backend 'nccl' (= RCCL): the data-parallel code path — bucket launches with ncclAvg, work.wait() stream
    semantics around replayed step graphs, the deferred optimiser graph with the generator forward running while the D
    gradients are "on the wire" — on the one GPU this box has.  With one rank the exchange is the identity, so the result
    must equal the plain trainer's bit for bit."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(q[1]), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    from rick_amd.dist import DataParallelGrads
    from rick_amd.models import Discriminator, Generator
    from rick_amd.synth import synth_reals, synth_state_dict
    from rick_amd.train import RickTrainer, TrainConfig
    from tests.shapes import discriminator_shapes, generator_shapes
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    dev = 'cuda:0'

    def build():
        g = Generator(size, 512, 8, channel_multiplier=2)
        d = Discriminator(size, channel_multiplier=2)
        g.load_state_dict(synth_state_dict(generator_shapes(size)), strict=False)
        d.load_state_dict(synth_state_dict(discriminator_shapes(size)), strict=False)
        return g.to(dev), d.to(dev)

    out = {}
    for mode in ('plain', 'rccl'):
        import random
        random.seed(3)
        torch.manual_seed(3)
        g, d = build()
        g_ema, d_ema = build()
        dp = DataParallelGrads(bucket_bytes=bucket_bytes, force=True) if mode == 'rccl' else None
        tr = RickTrainer(TrainConfig(size=size, batch=B, warmup_iter=0), g, d, g_ema, d_ema, dp=dp)
        if dp is not None:
            assert dp.active and dist.get_backend() == 'nccl' and len(dp._state[id(tr.d_flat)]['buckets']) >= 3
        tr.enable_graphs(True)
        reals = [synth_reals(B, size=size, seed=40 + k).to(dev) for k in range(3)]
        if direct_steps:
            # the step API used directly: consecutive graph G steps — the second one's head (generator
            # forward) must not run ahead of the first one's still pending optimiser graph, which updates G itself
            tr._real = reals[0].clone()
            for k in range(4):
                tr.d_step(tr._real, None, graph=True)
                tr.g_step(None, graph=True)
                tr.g_step(None, graph=True)
            if dp is not None:
                assert tr._pending is not None and tr._pending[1] is tr.g_flat
            tr.ema_step()
        else:
            for i in range(16, 16 + iters):              # i = 16: R1 + path length; captures happen on each step type's 3rd call
                tr.iteration(i, reals[i % 3])
        if dp is not None:
            assert tr._gs['g']['graphs'][0] is not None and tr._gs['g']['graphs'][2] is not None    # head | fwd/bwd | optimiser
            assert tr._pending is None                   # ema_step completed the deferred optimiser step
        torch.cuda.synchronize()
        out[mode] = [t.detach().clone() for t in (tr.g_flat.flat, tr.d_flat.flat, tr.g_ema_flat.flat, tr.d_ema_flat.flat)]
        out[mode].append(float(tr.losses['g']))
    res = {'buckets': (len(dp._state[id(tr.g_flat)]['buckets']), len(dp._state[id(tr.d_flat)]['buckets'])),
           'equal': [bool(torch.equal(a, b)) for a, b in zip(out['plain'][:4], out['rccl'][:4])],
           'finite': [bool(torch.isfinite(a).all()) for a in out['rccl'][:4]],
           'g_loss': (out['plain'][4], out['rccl'][4])}
    dist.destroy_process_group()
    q[0].put(res)


def test_rccl_code_path_single_rank_pipelined_graphs():
    """RCCL ('nccl' backend) has never seen two devices on this pool, but its code path can run with ONE rank: nine RICK
    iterations with step graphs through DataParallelGrads(force=True) — AVG all-reduces launched behind the replayed
    forward/backward graph, the optimiser graph deferred behind the next step's head — leave exactly the parameters, EMA
    weights and losses of the plain single-process trainer."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    p = ctx.Process(target=_nccl_worker, args=((q, port),))
    p.start()
    out = _get(q, [p], 600)
    p.join(60)
    assert p.exitcode == 0
    assert all(out['equal']) and all(out['finite']), out
    assert out['g_loss'][0] == out['g_loss'][1]


def test_rccl_code_path_single_rank_256px_batch4_default_buckets():
    """The same at BASELINE config 3's PER-RANK shapes (256 px, batch 4) with the default 32 MiB buckets: 3 collectives for the
    generator's gradients (95 MB) and 4 for the discriminator's behind the replayed 256-px step graphs, R1 and path-length
    steps included (i = 16), nine iterations; parameters, EMA weights and the last loss equal the plain trainer's bit for bit."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    p = ctx.Process(target=_nccl_worker, args=((q, port), False, 256, 4, 32 << 20, 9))
    p.start()
    out = _get(q, [p], 900)
    p.join(60)
    assert p.exitcode == 0
    assert out['buckets'][0] >= 3 and out['buckets'][1] >= 3, out['buckets']
    assert all(out['equal']) and all(out['finite']), out
    assert out['g_loss'][0] == out['g_loss'][1]


def test_rccl_consecutive_graph_g_steps_finish_their_own_pending_step():
    """Two graph G steps in a row through the step API under the pipelined data-parallel mode: the pending optimiser graph
    of the first updates G, so it has to land before the second step's head replays (train.py `_run`).  Parameters equal the
    plain trainer's bit for bit."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    p = ctx.Process(target=_nccl_worker, args=((q, port), True))
    p.start()
    out = _get(q, [p], 600)
    p.join(60)
    assert p.exitcode == 0
    assert all(out['equal']) and all(out['finite']), out
    assert out['g_loss'][0] == out['g_loss'][1]
