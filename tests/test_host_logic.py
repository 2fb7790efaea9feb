"""CPU-only tests of the host-side logic: Fisher decisions vs the oracle restatement and the
captured reference FIM, mask construction, flat parameter views, and the multi-process
data-parallel path (gloo, world_size 2)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle.train_ref import d_decisions_ref, g_decisions_ref, zero_idx_merge_ref
from rick_amd.train import (FlatParams, build_mask, d_optim_filter, decide_d, decide_g, g_optim_filter,
                            zero_idx_merge)
from tests.shapes import discriminator_shapes, generator_shapes


_FF = {}


def fake_fisher(shapes, seed):
    """Synthetic non-negative 'Fisher' tensors for the keys the decision code reads (cached)."""
    if seed not in _FF:
        rs = np.random.RandomState(seed)
        _FF[seed] = {k: (rs.rand(*s).astype(np.float32) ** 4) for k, s in shapes.items()
                     if k.startswith('convs.') and not k.endswith('.kernel')}
    return _FF[seed]


@pytest.mark.parametrize('fq,pq', [(40, 0.1), (85, 0.075), (75, 0.1)])
def test_decisions_match_oracle(fq, pq):
    """product decide_g / decide_d (fed per-filter vectors) == oracle restatement of
    train_dynamic_update_prune.py:279-393 (fed full Fisher tensors)."""
    fg = fake_fisher(generator_shapes(256), 1)
    fd = fake_fisher(discriminator_shapes(256), 2)
    conv = {f'convs.{k}.conv.weight': fg[f'convs.{k}.conv.weight'].mean(axis=(0, 2, 3, 4)) for k in range(12)}
    fc = {f'convs.{k}.conv.modulation.weight': (fg[f'convs.{k}.conv.modulation.weight'].mean(axis=1)
                                                 + fg[f'convs.{k}.conv.modulation.bias']) / 2 for k in range(12)}
    got = decide_g(conv, fc, fq, pq)
    ref = g_decisions_ref(fg, fq, pq)
    for a, b in zip(got, ref):
        assert a.keys() == b.keys()
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    dfim = {}
    for b in range(1, 7):
        for li in range(2):
            wk, bk = f'convs.{b}.conv{li + 1}.{li}.weight', f'convs.{b}.conv{li + 1}.{li + 1}.bias'
            dfim[wk] = (fd[wk].mean(axis=(1, 2, 3)) + fd[bk]) / 2
        dfim[f'convs.{b}.skip.1.weight'] = fd[f'convs.{b}.skip.1.weight'].mean(axis=(1, 2, 3))
    got = decide_d(dfim, fq, pq)
    ref = d_decisions_ref(fd, fq, pq)
    for a, b in zip(got, ref):
        assert a.keys() == b.keys()
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    # every filter lands in exactly one set; D bias keys exist (character arithmetic of :363)
    fr, ft, pr = got
    assert 'convs.3.conv2.2.bias' in fr and 'convs.3.conv1.1.bias' in fr
    for k in fr:
        n = len(fr[k]) + len(ft[k]) + len(pr[k])
        assert n == len(dfim[k if k in dfim else k.replace(f'{int(k[-6])}.bias', f'{int(k[-6]) - 1}.weight')])
    m = zero_idx_merge(pr, fr)
    mr = zero_idx_merge_ref(pr, fr)
    assert all(np.array_equal(m[k], mr[k]) for k in m)


def test_decisions_on_reference_fim(golden):
    """README quantiles on the FIM captured from the reference's estimate_fisher at 256 px."""
    g = golden('full256')
    conv = {f'convs.{k}.conv.weight': g[f'fisher/g_conv/{k}'] for k in range(12)}
    fc = {f'convs.{k}.conv.modulation.weight': g[f'fisher/g_fc/{k}'] for k in range(12)}
    for fq, pq in ((40, 0.1), (85, 0.075)):
        fr, ft, pr = decide_g(conv, fc, fq, pq)
        allc = np.concatenate([conv[k] for k in conv])
        cut, prl = np.percentile(allc, fq), np.percentile(allc, pq)
        for k, v in conv.items():
            assert np.array_equal(fr[k], np.where(v > cut)[0])
            assert np.array_equal(pr[k], np.where(v <= prl)[0])
        tot_freeze = sum(len(fr[k]) for k in conv)
        assert abs(tot_freeze - round(4864 * (100 - fq) / 100)) <= 2
        assert set(fr) == set(conv) | set(fc) | {k.replace('weight', 'bias') for k in fc}


@pytest.mark.parametrize('qtag,fq,pq', [('q40', 40.0, 0.1), ('q85', 85.0, 0.075)])
def test_decisions_equal_reference_block(golden, qtag, fq, pq):
    """Row Q pinned by the reference's OWN code: tests/golden/rick256.npz holds the index sets its decision block
    (train_dynamic_update_prune.py:277-393, executed as a source slice by tools/make_golden.py on Fisher dictionaries from
    its estimate_fisher, 256 px, two sweeps of two samples) produced for the README quantiles, next to the per-filter
    FIM vectors in its own expressions.  decide_g / decide_d / zero_idx_merge must reproduce every set EXACTLY — freeze,
    fine-tune and prune sets of both networks (incl. the >= / < rule of the skip layers and the bias keys derived by
    character arithmetic), the percentile lines, and the cumulative prune set after the second sweep."""
    g = golden('rick256')
    zero_g = zero_d = None
    for sw in range(2):
        conv = {k.split('/', 2)[2]: g[k] for k in g.files if k.startswith(f'fim{sw}/g_conv/')}
        fc = {k.split('/', 2)[2]: g[k] for k in g.files if k.startswith(f'fim{sw}/g_fc/')}
        dfim = {k.split('/', 2)[2]: g[k] for k in g.files if k.startswith(f'fim{sw}/d/')}
        assert len(conv) == 12 and len(fc) == 12 and len(dfim) == 18
        fr_g, ft_g, pr_g = decide_g(conv, fc, fq, pq)
        fr_d, ft_d, pr_d = decide_d(dfim, fq, pq)
        if sw == 0:                                          # :386-393
            zero_g, zero_d = pr_g, pr_d
        else:
            zero_g, zero_d = zero_idx_merge(zero_g, pr_g), zero_idx_merge(zero_d, pr_d)
        for name, got in (('idx_freeze_g', fr_g), ('idx_ft_g', ft_g), ('idx_prune_g', pr_g), ('idx_freeze_d', fr_d),
                          ('idx_ft_d', ft_d), ('idx_prune_d', pr_d), ('zero_filter_idx_g', zero_g),
                          ('zero_filter_idx_d', zero_d)):
            pre = f'{qtag}/s{sw}/{name}/'
            ref = {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}
            assert ref.keys() == got.keys(), (name, set(ref) ^ set(got))
            for k in ref:
                assert np.array_equal(np.asarray(got[k]), ref[k]), (name, k, sw)
        # the percentile lines themselves
        allc = np.concatenate([[]] + [conv[f'convs.{k}.conv.weight'] for k in range(12)], axis=None)
        assert np.percentile(allc, q=fq) == float(g[f'{qtag}/s{sw}/cutline_g_conv'])
        assert np.percentile(allc, q=pq) == float(g[f'{qtag}/s{sw}/pruneline_g_conv'])
    assert sum(len(v) for v in zero_g.values()) >= sum(len(v) for v in pr_g.values())


def test_flat_params_and_masks_cpu():
    from rick_amd.models import Generator
    g = Generator(16, 512, 8)
    named = [(n, p) for n, p in g.named_parameters() if g_optim_filter(n)]
    before = {n: p.detach().clone() for n, p in named}
    flat = FlatParams(named)
    from rick_amd.train import FLAT_ALIGN
    assert flat.total == sum(-(-p.numel() // FLAT_ALIGN) * FLAT_ALIGN for _, p in named)      # 256-byte aligned starts
    assert all(int(o) % FLAT_ALIGN == 0 for o in flat.offsets)
    for n, p in named:
        assert torch.equal(p.detach(), before[n])
        lo, hi = flat.segment(n)
        assert p.data_ptr() == flat.flat[lo:hi].data_ptr() and p.grad.data_ptr() == flat.grad[lo:hi].data_ptr()
    # autograd accumulates into the flat views
    sum((p * p).sum() for _, p in named).backward()
    assert torch.allclose(flat.grad, 2 * flat.flat.detach())
    flat.zero_grad()
    assert float(flat.grad.abs().max()) == 0
    freeze = {'convs.0.conv.weight': np.array([1, 3]), 'convs.0.conv.modulation.bias': np.array([0])}
    zero = {'convs.1.conv.weight': np.array([2]), 'not.a.key': np.array([1])}
    mask = build_mask(flat, freeze, zero)
    lo, hi = flat.segment('convs.0.conv.weight')
    mv = mask[lo:hi].view(1, 512, 512, 3, 3)
    assert int(mv[:, [1, 3]].min()) == 1 and int(mv[:, [0, 2, 4]].max()) == 0
    lo, hi = flat.segment('convs.1.conv.weight')
    assert int(mask[lo:hi].view(1, 512, 512, 3, 3)[:, 2].min()) == 2
    lo, hi = flat.segment('convs.0.conv.modulation.bias')
    assert mask[lo:hi].tolist()[:2] == [1, 0]
    assert d_optim_filter('convs.3.conv1.0.weight') and d_optim_filter('final_linear.1.bias')
    assert not d_optim_filter('convs.0.0.weight')


# ------------------------------------------------------------------- world_size-2 gloo
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from rick_amd.dist import DataParallelGrads, init_from_env
    init_from_env('gloo')
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(64, 300), torch.nn.Linear(300, 300), torch.nn.Linear(300, 7))
    named = list(lin.named_parameters())
    flat = FlatParams(named)
    dp = DataParallelGrads(bucket_bytes=64 * 1024)
    dp.broadcast_params([lin])
    dp.attach(flat)
    assert len(dp._state[id(flat)]["buckets"]) >= 2
    xs = torch.randn(8, 64, generator=torch.Generator().manual_seed(5))
    res = []
    for it in range(2):                       # two rounds: hooks re-arm
        flat.zero_grad()
        x = xs[rank * 4:(rank + 1) * 4]
        lin(x).pow(2).mean().backward()       # per-rank mean over its micro-batch
        dp.all_reduce(flat)
        res.append(flat.grad.clone())
    # frozen parameter (warm-up stage): its bucket must still complete
    named[0][1].requires_grad = False
    flat.zero_grad()
    dp.prepare(flat)
    lin(xs[rank * 4:(rank + 1) * 4]).pow(2).mean().backward()
    dp.all_reduce(flat)
    vec = [torch.full((5,), float(rank + 1)), torch.full((3,), 10.0 * (rank + 1))]
    dp.all_reduce_vectors(vec)
    def dense(gr):          # gradient of every parameter, concatenated (the flat buffer pads each one to 256 bytes)
        return torch.cat([gr[slice(*flat.segment(n))] for n in flat.names]).numpy()
    q.put((rank, dense(res[0]), dense(res[1]), vec[0].numpy(), vec[1].numpy(), dense(flat.grad.clone())))
    torch.distributed.destroy_process_group()


def test_data_parallel_grads_gloo_world2():
    """Averaged bucketed all-reduce == single-process gradient of the global-batch mean."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(64, 300), torch.nn.Linear(300, 300), torch.nn.Linear(300, 7))
    xs = torch.randn(8, 64, generator=torch.Generator().manual_seed(5))
    lin(xs).pow(2).mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in lin.parameters()]).numpy()
    for o in outs:
        assert np.allclose(o[1], ref, rtol=1e-5, atol=1e-7)
        assert np.allclose(o[2], ref, rtol=1e-5, atol=1e-7)
        assert np.allclose(o[3], 3.0) and np.allclose(o[4], 30.0)
        n0 = 64 * 300
        assert np.allclose(o[5][n0:], ref[n0:], rtol=1e-5, atol=1e-7)     # frozen first weight excluded
    assert np.array_equal(outs[0][1], outs[1][1])


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` as the driver calls it (no launcher, no WORLD_SIZE): the script starts its own two rank
    processes, they rendezvous on 127.0.0.1 (gloo here) and rank 0 prints ONE JSON line; exit code 0.  RICK_BENCH_DRYRUN
    stops each rank after the rendezvous + one all-reduce (this host has no GPU); the measured path of the same launcher
    is covered by tests/test_gpu_dp.py::test_bench_self_launch_two_ranks_one_gpu."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RICK_BENCH_DRYRUN='1', RICK_DIST_BACKEND='gloo')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {'dry_run': True, 'n_gpus': 2, 'rank_sum': 3.0, 'backend': 'gloo'}


def test_bench_self_launch_propagates_failure():
    """A rank that dies makes the launcher exit non-zero (and stops the surviving rank instead of leaving it in a
    collective)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RICK_BENCH_DRYRUN='1', RICK_DIST_BACKEND='no-such-backend')
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0


# ------------------------------------------------------------------- world_size-4 gloo
def _dp4_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from rick_amd.dist import DataParallelGrads, init_from_env
    init_from_env('gloo')
    # integer-valued data: every gradient and every partial sum is exact in fp32, so the result cannot depend on the order
    # in which a ring sums the four ranks — bucketed and blocking exchange must agree BIT FOR BIT
    gen = torch.Generator().manual_seed(11)
    def ints(*shape, k=2):
        return torch.randint(-k, k + 1, shape, generator=gen).float()
    import copy
    nets, flats, twins = [], [], []
    for widths in ((48, 160, 96, 96, 8), (40, 200, 120, 64, 64, 4)):     # "G": 3 buckets, "D": 4 buckets at 16 KiB
        # INDEPENDENT layers (each reads the input's first columns): gradients are sums of a few small integers
        lin = torch.nn.ModuleList([torch.nn.Linear(a, b) for a, b in zip(widths[:-1], widths[1:])])
        with torch.no_grad():
            for p in lin.parameters():
                p.copy_(ints(*p.shape))
        nets.append(lin)
        flats.append(FlatParams(list(lin.named_parameters())))
        twin = copy.deepcopy(lin)                                  # same parameters, no hooks: the blocking reference
        twins.append((twin, FlatParams(list(twin.named_parameters()))))
    dp = DataParallelGrads(bucket_bytes=16 * 1024)
    dp.attach(*flats)
    nb = [len(dp._state[id(f)]['buckets']) for f in flats]
    xs = [ints(16, 256, k=3), ints(16, 256, k=3)]
    out = {'nb': nb}
    for name, lin, flat, x in zip('gd', nets, flats, xs):
        flat.zero_grad()
        dp.prepare(flat)
        xr = x[rank * 4:(rank + 1) * 4]
        sum(((rank + 1) * m(xr[:, :m.in_features])).sum() for m in lin).backward()   # buckets leave from the hooks during backward
        dp.all_reduce(flat)
        twin, tflat = twins['gd'.index(name)]
        tflat.zero_grad()
        sum(((rank + 1) * m(xr[:, :m.in_features])).sum() for m in twin).backward()
        blocking = tflat.grad.clone()
        dist.all_reduce(blocking, op=dist.ReduceOp.SUM)
        blocking /= world
        out[name] = (flat.grad.clone().numpy(), blocking.numpy())
    # Fisher sweep with fewer samples than ranks' worth: 5 samples over 4 ranks (2 / 1 / 1 / 1), then 3 samples (rank 3 has
    # none: its loop body never runs and its accumulators stay zero, train.py fisher_sweep) — the per-filter vectors are summed
    # across ranks and every rank must take identical decisions
    from rick_amd.train import decide_d
    for nsamp in (5, 3):
        rs = np.random.RandomState(100 + nsamp)
        per_sample = [{f'convs.{b}.{nm}': rs.rand(64).astype(np.float32) ** 3 for b in range(1, 7)
                       for nm in ('conv1.0.weight', 'conv2.1.weight', 'skip.1.weight')} for _ in range(nsamp)]
        mine = [j for j in range(nsamp) if j % world == rank]
        keys = sorted(per_sample[0])
        vecs = [torch.zeros(64) for _ in keys]
        for j in mine:
            for v, k in zip(vecs, keys):
                v += torch.from_numpy(per_sample[j][k])
        dp.all_reduce_vectors(vecs)
        fim = {k: v.numpy() / nsamp for k, v in zip(keys, vecs)}
        fr, ft, pr = decide_d(fim, 75, 0.1)
        out[f'fisher{nsamp}'] = (len(mine), {k: fim[k].copy() for k in keys}, {k: np.asarray(v) for k, v in fr.items()},
                                 {k: np.asarray(v) for k, v in pr.items()})
    q.put((rank, out))
    dist.destroy_process_group()


def test_data_parallel_world4_buckets_and_uneven_fisher_shards():
    """Four ranks (gloo, CPU): 3 + 4 buckets, bucketed == blocking all-reduce bit for bit (integer-valued gradients), equal on
    every rank; Fisher shards of 2/1/1/1 and 1/1/1/0 samples give every rank the same per-filter statistics and decisions."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp4_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert outs[0]['nb'] == [3, 4], outs[0]['nb']
    for name in 'gd':
        for r in range(4):
            got, blocking = outs[r][name]
            assert np.array_equal(got, blocking), (name, r)
            assert np.array_equal(got, outs[0][name][0])
        assert float(np.abs(outs[0][name][0]).max()) > 0
    for nsamp, counts in ((5, [2, 1, 1, 1]), (3, [1, 1, 1, 0])):
        assert [outs[r][f'fisher{nsamp}'][0] for r in range(4)] == counts
        ref = outs[0][f'fisher{nsamp}']
        for r in range(1, 4):
            o = outs[r][f'fisher{nsamp}']
            for k in ref[1]:
                assert np.array_equal(o[1][k], ref[1][k])
            assert o[2].keys() == ref[2].keys() and all(np.array_equal(o[2][k], ref[2][k]) for k in ref[2])
            assert o[3].keys() == ref[3].keys() and all(np.array_equal(o[3][k], ref[3][k]) for k in ref[3])


def test_hand_over_attributes_under_inference_mode():
    """op/split.py hand() / taken() read tensor._version, which inference tensors do not have: under
    torch.inference_mode() nothing is attached and nothing is returned — the consumer measures again — instead of a crash."""
    import torch
    from rick_amd.op import split as sp
    with torch.inference_mode():
        t = torch.zeros(4)
        sp.hand(t, '_rick_amax', 1.0)
        assert sp.taken(t, '_rick_amax') is None
        u = torch.ones(4)
        sp.rehand(t, u)
        assert sp.taken(u, '_rick_amax') is None
    v = torch.zeros(4)
    sp.hand(v, '_rick_amax', 2.0)
    assert sp.taken(v, '_rick_amax') == 2.0
    v.add_(1)
    assert sp.taken(v, '_rick_amax') is None
