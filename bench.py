#!/usr/bin/env python3
"""Headline benchmark: G+D train-step images/sec @256 px (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under a launcher — python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
     — or bare: without WORLD_SIZE in the environment the script starts its own N rank processes, one per GPU, before
     anything touches the GPU, waits for them and exits non-zero if any rank failed)

A "step" is one iteration of the RICK adaptation loop in steady state (i >= warmup_iter,
train_dynamic_update_prune.py:395-589,697-698): D step, [R1 every 16], G step, [path length
every 4], EMA — with the freeze/prune masks of a Fisher sweep active — on the configuration the
metric is quoted on (BASELINE configs[1]: FFHQ-256 architecture, batch 4 per GPU).  Weights are
random-init (torch.manual_seed(1); the FFHQ checkpoint is a download), reals are seeded U(-1,1)
batches already resident in HBM.  value = global batch * K / max-over-ranks wall time.

Extra objects on the JSON line:
  roofline      the dominant kernel (conv_igemm_kernel, fp16x3 MFMA): algorithmic FLOPs / time,
                timed per launch with HIP events on the launch stream in an instrumented repeat of
                the same steps (the timed region itself carries no events)
  cpu_baseline  the CPU oracle (port of the reference's CPU formulation) timed on this box's host
                cores on a bounded sample (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0      # /opt/skills/guides/MI355X_MICROARCH.md
HBM_LARGE_BYTES = 16 << 20


def cpu_baseline(size=256, batch=2, timed=2):
    """Non-reg iterations (D step fwd+bwd, G step fwd+bwd) of the CPU oracle at BASELINE configs[0]'s batch 2: one warm-up
    iteration (oneDNN primitive creation, allocator), then `timed` timed ones.  Thread count: the box has 256
    hardware threads, but oneDNN's convolutions at this size collapse beyond one socket's worth — measured on the GPU box:
    12 s per iteration at 32 threads, 335 s at 256 (gpurun_out/r2_bench_full.log) — so the baseline runs at the count
    that is FASTEST for the CPU path (32, or every core of a smaller host) and reports both numbers."""
    from oracle.model_ref import discriminator_ref, generator_ref
    from oracle.train_ref import d_logistic_loss_ref, g_nonsaturating_loss_ref
    from rick_amd.synth import synth_latents, synth_reals, synth_state_dict
    from tests.shapes import discriminator_shapes, generator_shapes
    host = os.cpu_count() or 1
    cores = min(host, 32)
    torch.set_num_threads(cores)
    sg = synth_state_dict(generator_shapes(size))
    sd = synth_state_dict(discriminator_shapes(size))
    pg = [v.requires_grad_(True) for k, v in sg.items() if not k.startswith('noises.')]
    pd = [v.requires_grad_(True) for k, v in sd.items()]
    z, real = synth_latents(batch, seed=11), synth_reals(batch, size=size, seed=11)

    def iteration():
        with torch.no_grad():
            fake, _ = generator_ref(sg, [z], size=size)
        fp, _ = discriminator_ref(sd, fake, size=size)
        rp, _ = discriminator_ref(sd, real, size=size)
        torch.autograd.grad(d_logistic_loss_ref(rp, fp), pd)
        fake, _ = generator_ref(sg, [z], size=size)
        fp, _ = discriminator_ref(sd, fake, size=size)
        torch.autograd.grad(g_nonsaturating_loss_ref(fp), pg, allow_unused=True)

    iteration()
    times = []
    for _ in range(timed):
        t0 = time.perf_counter()
        iteration()
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2] if len(times) % 2 else sum(times) / len(times)
    return {'value': batch / dt, 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'1 warm-up + {timed} timed non-reg iterations (D step + G step, fwd+bwd, no optimiser) at batch '
                      f'{batch}, {size}px, fp32 oneDNN on {cores} of {host} host threads (fastest setting for the CPU path; 256 '
                      f'threads measured 28x slower): ' + ', '.join(f'{t:.1f}' for t in times) + ' s'}


def kernel_source_hash():
    """sha256 over rick_amd/csrc/*.{hip,h} (tools/pmc_traffic.py stores the same digest with the counters)."""
    import hashlib
    root = os.path.join(ROOT, 'rick_amd', 'csrc')
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith(('.hip', '.h')):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), 'rb').read())
    return h.hexdigest()[:16]


def load_traffic():
    """(HBM bytes per conv_igemm launch, provenance) from the newest committed rocprofv3 --pmc passes
    (profiles/rNN_pmc_traffic.json, produced by tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE passes over the
    bench iteration, FETCH_SIZE doubled as the guide prescribes for 16-B/lane reads on gfx950).  The provenance says
    whether the kernel sources are still the ones the counters were collected with.  (None, None) when no file exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    sha = d.get('kernel_source_sha16')
    return d.get('conv_igemm', {}).get('hbm_bytes_per_launch'), {
        'file': os.path.relpath(files[-1], ROOT), 'kernel_source_sha16': sha, 'current_kernel_source_sha16': kernel_source_hash(),
        'counters_describe_current_kernels': sha == kernel_source_hash()}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, rendezvous on 127.0.0.1), one per GPU, as CHILDREN of this process — which has not touched the GPU and
    never will (no exec of a GPU-initialised process) — let rank 0 print the JSON line on the inherited stdout, and return
    the first non-zero exit code (the other ranks are terminated: a lost rank would leave them in a collective)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f'[bench] rank {r} exited with code {code}; stopping the other ranks', file=sys.stderr)
                for q in live:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=32)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--batch', type=int, default=4, help='per-GPU batch (BASELINE configs[1])')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--precision', default='fp16x3', choices=['fp16x3', 'fp16'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-step-times', action='store_true')
    ap.add_argument('--no-fisher', action='store_true', help='skip the (untimed) Fisher sweep; masks stay empty (profiling runs)')
    ap.add_argument('--no-graphs', action='store_true', help='issue every launch from Python instead of replaying captured hipGraphs')
    ap.add_argument('--graphs', action='store_true', help='(default) replay captured step graphs; with N > 1 each step is a '
                    'forward/backward graph, the bucketed RCCL all-reduce, and an optimiser graph (--no-graphs: eager issue, '
                    'all-reduce launched from autograd hooks and overlapped with backward, but host-bound)')
    ap.add_argument('--eval', action='store_true', help='(default on) also time G inference (BASELINE config 4: batches of 25)')
    ap.add_argument('--no-extras', action='store_true', help='skip the untimed extras: 64-sample Fisher sweep (config 5), G inference (config 4)')
    ap.add_argument('--fisher-img', type=int, default=2, help='samples of the (untimed) Fisher sweep that sets the masks')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus))

    from rick_amd import op
    from rick_amd.dist import DataParallelGrads, init_from_env
    from rick_amd.models import Discriminator, Generator
    from rick_amd.op.conv import launch_profiler
    from rick_amd.synth import synth_latents, synth_reals
    from rick_amd.train import RickTrainer, TrainConfig
    import torch.distributed as dist

    # stdout carries exactly ONE line, the JSON: everything libraries write to fd 1 through C stdio (RCCL prints a version
    # banner on rank 0) goes to stderr instead; the line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        os.write(json_fd, (json.dumps(obj) + '\n').encode())

    rank, local, world = init_from_env()
    if os.environ.get('RICK_BENCH_DRYRUN'):
        # launcher check for hosts without a GPU (tests/test_host_logic.py): rendezvous, one all-reduce over the ranks,
        # rank 0 prints a line; no kernel runs and nothing is measured
        t = torch.tensor([float(rank + 1)])
        if world > 1:
            dist.all_reduce(t)
            dist.barrier()
        if rank == 0:
            emit({'dry_run': True, 'n_gpus': world, 'rank_sum': float(t), 'backend': dist.get_backend() if world > 1 else None})
        if world > 1:
            dist.destroy_process_group()
        return
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    if os.environ.get('RICK_FORCE_DEVICE') is not None:      # functional test: several ranks on one GPU (gloo)
        local = int(os.environ['RICK_FORCE_DEVICE'])
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    op.set_precision(args.precision)

    torch.manual_seed(1)                                   # train_dynamic_update_prune.py:760-762
    cfg = TrainConfig(size=args.size, batch=args.batch, num_fisher_img=args.fisher_img)
    g = Generator(cfg.size, cfg.latent, cfg.n_mlp, cfg.channel_multiplier).to(dev)
    d = Discriminator(cfg.size, cfg.channel_multiplier).to(dev)
    g_ema = Generator(cfg.size, cfg.latent, cfg.n_mlp, cfg.channel_multiplier).to(dev)
    d_ema = Discriminator(cfg.size, cfg.channel_multiplier).to(dev)
    g_ema.load_state_dict(g.state_dict())
    d_ema.load_state_dict(d.state_dict())
    dp = DataParallelGrads() if world > 1 else None
    if dp is not None:
        dp.broadcast_params([g, d, g_ema, d_ema])
    tr = RickTrainer(cfg, g, d, g_ema, d_ema, dp=dp)
    torch.manual_seed(1234 + rank)                         # per-rank latents / noise

    reals = [synth_reals(cfg.batch, cfg.size, seed=100 * rank + j).to(dev) for j in range(4)]
    # steady state: a Fisher sweep has run, masks are active (untimed; it recurs every fisher_freq iterations)
    mine = [j for j in range(cfg.num_fisher_img) if j % world == rank]
    fisher_in = ([synth_latents(1, seed=500 + j).to(dev) for j in mine], [synth_reals(1, cfg.size, seed=600 + j).to(dev) for j in mine])
    if not args.no_fisher:
        tr.fisher_sweep(*fisher_in, first=True)

    i0 = cfg.warmup_iter + 1
    use_graphs = not args.no_graphs
    if use_graphs:
        try:
            tr.enable_graphs(True)
            tr.prepare_graphs(reals[0])                    # untimed, like the Fisher sweep: steady state = graphs captured
        except Exception as e:                             # noqa: BLE001 — a capture problem must not cost the measurement
            print(f'[bench] hipGraph capture failed ({type(e).__name__}: {e}); issuing launches eagerly', file=sys.stderr)
            use_graphs = False
            tr.enable_graphs(False)

    fisher_ms = None
    if not args.no_fisher and use_graphs:
        # the sweep itself (untimed in `value`: it recurs every fisher_freq = 50 iterations): once more to capture its
        # per-sample graph, then timed.  Every rank runs it (its per-filter all-reduce must stay matched).
        tr.fisher_sweep(*fisher_in, first=True)
        torch.cuda.synchronize()
        tf = time.perf_counter()
        tr.fisher_sweep(*fisher_in, first=True)
        torch.cuda.synchronize()
        fisher_ms = 1e3 * (time.perf_counter() - tf)

    def run(n, start):
        for k in range(n):
            tr.iteration(start + k, reals[k % len(reals)])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    run(args.warmup, i0)
    fence()
    t0 = time.perf_counter()
    run(args.steps, i0 + args.warmup)
    fence()
    elapsed = time.perf_counter() - t0
    rank_ms = [1e3 * elapsed / args.steps]
    if world > 1:
        t = torch.zeros(world, device=dev if dist.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
        t[rank] = elapsed
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rank_ms = [1e3 * float(v) / args.steps for v in t]
        elapsed = float(t.max())

    out = {
        'metric': 'G+D train-step images/sec @256px', 'value': cfg.batch * world * args.steps / elapsed,
        'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32 (fp16x3 split MFMA, fp32 accumulate)' if args.precision == 'fp16x3' else 'f16',
        'data': 'synthetic',
        'config': {'workload': f'FFHQ-256 StyleGAN2 G+D RICK iteration (D step, R1/16, G step, PLR/4, EMA, masks on), '
                               f'batch {cfg.batch}/GPU, {cfg.size}px, channel_multiplier 2, random-init weights',
                   'global_batch': cfg.batch * world, 'parallelism': f'dp{world}', 'hip_graphs': use_graphs,
                   'first_iteration': i0 + args.warmup},
        'ranks': {'backend': dist.get_backend() if world > 1 else None, 'world_size': world,
                  'rccl_ranks': world if world > 1 and dist.get_backend() == 'nccl' else (0 if world > 1 else None),
                  'ms_per_step_per_rank': rank_ms},
    }

    if fisher_ms is not None:
        out['fisher_sweep'] = {'ms': fisher_ms, 'samples_this_rank': len(mine), 'samples': cfg.num_fisher_img,
                               'note': 'whole sweep (per-sample G/D forward + both gradients at batch 1 replayed from a captured '
                                       'graph, grad^2 accumulate, per-filter reduce, percentile decisions, mask upload); not part '
                                       'of `value`; the README recipe runs 5 samples every 50 iterations'}

    # ---- per-step-type times (untimed extra pass, every rank runs it so collectives stay matched): HIP events on the
    # launch stream after every step of 32 more iterations = two full R1 periods (8 path-length steps, 2 R1 steps, 24 iterations
    # without a regulariser: SURVEY 8d's headline is the median of >= 20 of those)
    if not args.no_step_times:
        tr.step_events = []
        run(32, i0 + args.warmup + args.steps)
        torch.cuda.synchronize()
        ev, tr.step_events = tr.step_events, None
        per, iters, cur, t_begin = {}, [], [], None
        for (name, e), (_, prev) in zip(ev[1:], ev[:-1]):
            if name == 'begin':
                continue
            per.setdefault(name, []).append(prev.elapsed_time(e))
        for name, e in ev:
            if name == 'begin':
                t_begin, cur = e, []
            else:
                cur.append(name)
                if name == 'ema':
                    iters.append((t_begin.elapsed_time(e), tuple(cur)))
        med = lambda v: sorted(v)[len(v) // 2]                                         # noqa: E731
        nonreg = [t for t, names in iters if 'r1' not in names and 'plr' not in names]
        out['step_ms'] = {k: med(v) for k, v in per.items()}
        if world > 1 and use_graphs:
            out['step_ms_note'] = ('pipelined: with step graphs under data parallelism a step\'s all-reduce wait and optimiser '
                                   'graph run behind the NEXT step\'s head and are charged to that step')
        out['nonreg_iteration'] = {'median_ms': med(nonreg), 'images_per_s': cfg.batch * world / (1e-3 * med(nonreg)),
                                   'n': len(nonreg), 'note': 'D step + G step + EMA (SURVEY 8d definition), GPU time of this rank'}
        tot16 = sum(t for t, _ in iters)
        out['amortised16'] = {'ms_per_iteration': tot16 / len(iters), 'images_per_s': cfg.batch * world * len(iters) / (1e-3 * tot16),
                              'note': f'{len(iters)} consecutive iterations; per 16: 16 D + 16 G + 1 R1 + 4 path-length steps'}

    if not args.no_roofline:
        # instrumented repeat of 16 iterations: HIP events around every conv-family launch (eager issue; every rank
        # runs it — the bucketed all-reduces must stay matched — rank 0 reports)
        tr.enable_graphs(False)
        with launch_profiler() as prof:
            run(16, i0 + args.warmup + args.steps + 32)
            torch.cuda.synchronize()
        agg = {}
        by_tag = {}
        hbm = {}
        for kind, flops, e0, e1, tag, abytes in prof:
            dt = e0.elapsed_time(e1) * 1e-3
            if kind == 'hbm':           # HBM-bound launches (op.conv.hbm_launch): tag = kernel family, abytes = bytes to move once
                h = hbm.setdefault(tag, [0.0, 0.0, 0, 0.0, 0.0, 0])
                h[0] += abytes
                h[1] += dt
                h[2] += 1
                if abytes >= HBM_LARGE_BYTES:       # launches long enough that the two bracketing events do not weigh
                    h[3] += abytes
                    h[4] += dt
                    h[5] += 1
                continue
            a = agg.setdefault(kind, [0.0, 0.0, 0, 0.0])
            a[0] += flops
            a[1] += dt
            a[2] += 1
            a[3] += abytes
            bt = by_tag.setdefault(tag, [0.0, 0.0, 0])
            bt[0] += flops
            bt[1] += dt
            bt[2] += 1
        if os.environ.get('RICK_BENCH_BREAKDOWN') and rank == 0:
            for tag, (fl, dt, n) in sorted(by_tag.items(), key=lambda kv: -kv[1][1]):
                print(f'  {tag:44s} n={n:4d} total {dt*1e3/16:7.3f} ms/step  avg {dt/n*1e6:8.1f} us  {fl/dt/1e12:6.1f} TF',
                      file=sys.stderr)
        ig = agg.get('igemm', [0.0, 1.0, 1, 0.0])
        traffic, traffic_src = load_traffic()
        ach = ig[0] / ig[1] / 1e12
        mult = 3.0 if args.precision == 'fp16x3' else 1.0
        out['roofline'] = {'bound': 'mfma', 'kernel': 'conv_igemm_kernel', 'achieved': ach,
                           'peak': MFMA_BF16_DENSE_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                           'frac': ach / MFMA_BF16_DENSE_PEAK_TFLOPS, 'traffic': traffic, 'traffic_source': traffic_src,
                           'algorithmic_bytes_per_launch': ig[3] / max(ig[2], 1),
                           'mfma_issued_frac': mult * ach / MFMA_BF16_DENSE_PEAK_TFLOPS,
                           'launches': ig[2], 'avg_launch_us': 1e6 * ig[1] / max(ig[2], 1),
                           'algorithmic_gflop_per_launch': ig[0] / max(ig[2], 1) / 1e9,
                           'note': 'achieved = algorithmic FLOPs (2*N*OH*OW*Co*Ci*taps) / event-timed duration over '
                                   '16 instrumented iterations (forward, data-gradient and transposed launches of the '
                                   'igemm family incl. their split-K second stage); fp16x3 issues 3 MFMA FLOPs per '
                                   'algorithmic FLOP; traffic = HBM bytes per launch from the committed counter passes (traffic_source says '
                                   'whether they were taken with the current kernel sources); algorithmic_bytes_per_launch = fp32 input + '
                                   'weights + output of each launch, measured in this run'}
        if 'wgrad' in agg:
            wg = agg['wgrad']
            out['roofline']['wgrad_kernel'] = {'achieved': wg[0] / wg[1] / 1e12, 'launches': wg[2],
                                               'avg_launch_us': 1e6 * wg[1] / max(wg[2], 1)}
        # one entry per kernel VARIANT of the MFMA family (own algorithmic FLOPs, own event time), the operand form
        # (split image / fp32 split on the fly) kept apart, so every fraction can be recomputed from profiles/ kernel by kernel
        import re as _re

        def variant(tag):
            m = _re.match(r'(conv|convT|wgrad) (\d+)(?:->|x)(\d+) k(\d) s(\d) N(\d+) (?:a)?(\d+)x', tag)
            if not m:
                return None
            op_, k, s_, res = m.group(1), int(m.group(4)), int(m.group(5)), int(m.group(7))
            form = 'split image' if tag.endswith('[split]') else 'fp32 operands'
            if op_ == 'wgrad':
                name = f'conv_wgrad_kernel {k}x{k} stride {s_}'
            elif k == 3 and s_ == 2 and op_ == 'convT':
                name = 'convt2_kernel (3x3 stride-2 transposed, one staging)'
            elif k == 3:
                name = f'conv_igemm_kernel 3x3 stride {s_}' + (' (dgrad)' if op_ == 'convT' else '')
            else:
                name = f'conv_igemm_kernel {k}x{k}'
            if res < 32 and op_ != 'wgrad':
                name += ', < 32^2 (split-K / small-patch forms)'
            return name, form
        kv = {}
        for tag, (fl, dt, n) in by_tag.items():
            v = variant(tag)
            if v is not None:
                a = kv.setdefault(v, [0.0, 0.0, 0])
                a[0] += fl
                a[1] += dt
                a[2] += n
        out['roofline']['kernels'] = [
            {'kernel': name, 'operands': form, 'achieved': fl / dt / 1e12, 'frac': fl / dt / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS,
             'launches': n, 'avg_launch_us': 1e6 * dt / n, 'ms_per_step': 1e3 * dt / 16, 'algorithmic_gflop_per_launch': fl / n / 1e9}
            for (name, form), (fl, dt, n) in sorted(kv.items(), key=lambda kv_: -kv_[1][1])]
        out['hbm_kernels'] = [
            {'kernel': name, 'achieved': by / dt / 1e9, 'unit': 'GB/s', 'peak': 8000.0, 'frac': by / dt / 8e12, 'launches': n,
             'ms_per_step': 1e3 * dt / 16, 'avg_launch_us': 1e6 * dt / n,
             # the same over the launches that move >= 16 MB: an event-bracketed 4 us launch reads 8-10 us (the < 16 MB launches
             # of these families take 3-11 us on the device, profiles/r05_hbm_microbench.txt, and are latency-, not HBM-bound)
             'large': ({'achieved': lby / ldt / 1e9, 'frac': lby / ldt / 8e12, 'launches': ln, 'ms_per_step': 1e3 * ldt / 16,
                        'min_bytes': HBM_LARGE_BYTES} if ln else None)}
            for name, (by, dt, n, lby, ldt, ln) in sorted(hbm.items(), key=lambda kv_: -kv_[1][1])]
        conv_s = sum(a[1] for a in agg.values()) / 16
        out['roofline']['conv_family_ms_per_step'] = 1e3 * conv_s
    if rank == 0 and not args.no_extras:
        # BASELINE config 5 at its own size: a 64-sample Fisher sweep (10 shipped-latent stand-ins + 54 seeded), this rank's
        # share, from the captured per-sample graph
        from rick_amd.synth import synth_latents as _sl, synth_reals as _sr
        tr.enable_graphs(use_graphs)
        n64 = 64
        cfg64 = cfg.num_fisher_img
        cfg.num_fisher_img = n64
        fin = ([_sl(1, seed=500 + j).to(dev) for j in range(n64)], [_sr(1, cfg.size, seed=600 + (j % 8)).to(dev) for j in range(n64)])
        if world == 1:
            tr.fisher_sweep(fin[0][:2], fin[1][:2], first=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr.fisher_sweep(*fin, first=True)
            torch.cuda.synchronize()
            out['fisher_sweep_64'] = {'ms': 1e3 * (time.perf_counter() - t0), 'samples': n64,
                                      'note': 'BASELINE config 5 (num_fisher_img = 64) on one GPU: per-sample graph replay, grad^2 '
                                              'accumulate, per-filter reduce, decisions, mask upload'}
        cfg.num_fisher_img = cfg64
    if rank == 0 and (args.eval or not args.no_extras):
        from rick_amd.evaluate import sample_images
        n = 500
        sample_images(g_ema, 50, 25)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sample_images(g_ema, n, 25)
        torch.cuda.synchronize()
        out['eval_sampling'] = {'images_per_s': n / (time.perf_counter() - t0), 'batch': 25, 'images': n,
                                'note': 'g_ema inference loop of gan_training/eval.py:34-41, images kept on device'}
    if rank == 0 and world == 1 and not args.no_extras and use_graphs:
        # The data-parallel code path on ONE GPU: the same loop through DataParallelGrads(force=True) on a single-rank RCCL
        # ('nccl') group — forward/backward graph, ncclAvg per 32 MiB bucket on RCCL's stream, optimiser graph deferred behind
        # the next step's head.  With one rank the exchange is the identity, so the ratio to the plain loop is the pipeline's
        # non-wire overhead — the only part of N-GPU efficiency that one GPU can measure.
        try:
            import socket
            s_ = socket.socket()
            s_.bind(('127.0.0.1', 0))
            port = s_.getsockname()[1]
            s_.close()
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
            import datetime
            dist.init_process_group('nccl', rank=0, world_size=1, timeout=datetime.timedelta(seconds=90))   # (an extra must not stall the line)
            torch.manual_seed(1)
            g2 = Generator(cfg.size, cfg.latent, cfg.n_mlp, cfg.channel_multiplier).to(dev)
            d2 = Discriminator(cfg.size, cfg.channel_multiplier).to(dev)
            g2e = Generator(cfg.size, cfg.latent, cfg.n_mlp, cfg.channel_multiplier).to(dev)
            d2e = Discriminator(cfg.size, cfg.channel_multiplier).to(dev)
            dp1 = DataParallelGrads(force=True)
            tr2 = RickTrainer(cfg, g2, d2, g2e, d2e, dp=dp1)
            if not args.no_fisher:
                tr2.fisher_sweep(*fisher_in, first=True)
            tr2.enable_graphs(True)
            tr2.prepare_graphs(reals[0])

            def timed(trn, n):
                for k in range(args.warmup):
                    trn.iteration(i0 + k, reals[k % len(reals)])
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for k in range(n):
                    trn.iteration(i0 + args.warmup + k, reals[k % len(reals)])
                torch.cuda.synchronize()
                return (time.perf_counter() - t1) / n
            tr.enable_graphs(True)
            plain_s = timed(tr, args.steps)
            forced_s = timed(tr2, args.steps)
            out['dp1_forced'] = {'images_per_s': cfg.batch / forced_s, 'plain_images_per_s': cfg.batch / plain_s, 'ratio': plain_s / forced_s,
                                 'buckets': {'g': len(dp1._state[id(tr2.g_flat)]['buckets']), 'd': len(dp1._state[id(tr2.d_flat)]['buckets'])},
                                 'bucket_mib': 32,
                                 'note': 'same loop through DataParallelGrads(force=True) on a single-rank RCCL group (collectives, stream '
                                         'waits, three graphs per step, deferred optimiser step), timed back to back with the plain loop'}
            dist.destroy_process_group()
        except Exception as e:                             # noqa: BLE001 — an extra must not cost the measurement
            out['dp1_forced'] = {'error': f'{type(e).__name__}: {e}'}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(cfg.size)
    if rank == 0:
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
