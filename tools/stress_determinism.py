"""Stress test for run-to-run determinism of the eager, sink-off trainer steps (the hook-driven data-parallel mode):
the six-step sequence of tests/test_gpu_dp.py is repeated REPS times from the same state inside one process — optionally
with `--procs N` copies of this process sharing the GPU (two ranks share cuda:0 in that test) — and after every step the flat
gradient / parameter buffers are compared ON THE DEVICE with the first repetition's.  Prints, per step and parameter, how often and by
how much a repetition differed.

    python tools/stress_determinism.py [--reps 60] [--procs 2] [--size 32] [--sync]"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(args):
    import torch
    from tools.diag_determinism import _LocalDP
    from rick_amd import op
    from rick_amd.models import Discriminator, Generator
    from rick_amd.synth import synth_latents, synth_reals, synth_state_dict, synth_tensor
    from rick_amd.train import RickTrainer, TrainConfig
    from tests.shapes import discriminator_shapes, generator_shapes
    size, B, dev = args.size, 2, 'cuda:0'

    def build():
        g = Generator(size, 512, 8, channel_multiplier=2)
        d = Discriminator(size, channel_multiplier=2)
        g.load_state_dict(synth_state_dict(generator_shapes(size)), strict=False)
        d.load_state_dict(synth_state_dict(discriminator_shapes(size)), strict=False)
        return g.to(dev), d.to(dev)
    g, d = build()
    tr = RickTrainer(TrainConfig(size=size, batch=B, warmup_iter=1), g, d, *build(), dp=_LocalDP('sync' if args.sync else 'plain'))
    z = synth_latents(B, seed=100).to(dev)
    real = synth_reals(B, size=size, seed=200).to(dev)
    noises = [synth_tensor(f'dpnoise/{i}', tuple(getattr(g.noises, f'noise_{i}').shape)).to(dev) for i in range(g.num_layers)]
    pl_noise = synth_tensor('dp/pl0', (1, 3, size, size)).to(dev)
    init = {'g': tr.g_flat.flat.clone(), 'd': tr.d_flat.flat.clone()}

    def reset():
        for nm, fp, opt in (('g', tr.g_flat, tr.g_optim), ('d', tr.d_flat, tr.d_optim)):
            fp.flat.copy_(init[nm])
            fp.grad.zero_()
            opt.m.zero_()
            opt.v.zero_()
            opt.steps[:] = [0] * len(opt.steps)
            opt.sync_steps_to_device()
        tr.mean_path_length = 0
        op.bump_weights_epoch()

    steps = [('0_d_warm', lambda: tr.d_step(real, [z], i=0, g_noise=noises)),
             ('1_r1_warm', lambda: tr.r1_step(real, i=0)),
             ('2_d', lambda: tr.d_step(real, [z], i=1, g_noise=noises)),
             ('3_g', lambda: tr.g_step([z], g_noise=noises)),
             ('4_plr', lambda: tr.plr_step([z[:1]], pl_noise=pl_noise, g_noise=noises)),
             ('5_d', lambda: tr.d_step(real, [z], i=2, g_noise=noises)),
             ('6_r1', lambda: tr.r1_step(real, i=16))]
    ref, events = {}, []
    for rep in range(args.reps):
        reset()
        for tag, fn in steps:
            fn()
            for nm, fp in (('g', tr.g_flat), ('d', tr.d_flat)):
                for kind, buf in (('grad', fp.grad), ('flat', fp.flat)):
                    key = f'{tag}/{nm}/{kind}'
                    if rep == 0:
                        ref[key] = buf.clone()
                    elif not torch.equal(buf, ref[key]):
                        diff = (buf != ref[key])
                        names = []
                        for n in fp.names:
                            lo, hi = fp.segment(n)
                            c = int(diff[lo:hi].sum())
                            if c:
                                names.append(f'{n}: {c}/{hi - lo} max|d| {float((buf[lo:hi] - ref[key][lo:hi]).abs().max()):.2e}')
                        events.append((rep, key, names))
            if events and events[-1][0] == rep:
                break                      # later steps of this repetition inherit the difference
    print(f'[pid {os.getpid()}] {args.reps} repetitions, {len({e[0] for e in events})} differed from the first', flush=True)
    for rep, key, names in events[:12]:
        print(f'   rep {rep} {key}:', '; '.join(names[:6]), '...' if len(names) > 6 else '', flush=True)
    return len(events)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=60)
    ap.add_argument('--procs', type=int, default=2)
    ap.add_argument('--size', type=int, default=32)
    ap.add_argument('--sync', action='store_true')
    ap.add_argument('--worker', action='store_true')
    args = ap.parse_args()
    if args.worker:
        sys.exit(1 if worker(args) else 0)
    cmd = [sys.executable, os.path.abspath(__file__), '--worker', '--reps', str(args.reps), '--size', str(args.size)] + (['--sync'] if args.sync else [])
    procs = [subprocess.Popen(cmd) for _ in range(args.procs)]       # (this parent never touches the GPU)
    rcs = [p.wait() for p in procs]
    print('exit codes', rcs)
    sys.exit(max(rcs))
