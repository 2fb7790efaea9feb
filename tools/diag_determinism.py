"""Run-to-run determinism of the eager, sink-off trainer steps (the mode the hook-driven data-parallel exchange uses).

    python tools/diag_determinism.py            # driver: runs the worker 2 x per mode in child processes and compares
    python tools/diag_determinism.py worker <mode> <out.npz>

Modes: 'plain' (hooks fire, do nothing), 'sync' (every hook calls torch.cuda.synchronize(): what the host-staged bucket
launch of tests/test_gpu_dp.py changes), 'copy' (every hook copies its parameter's gradient to the host).
After every step the flat gradients and parameters are snapshotted; the driver prints, per step and parameter, the
number of differing elements and the largest |difference|."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _LocalDP(mode):
    """Single-process stand-in for DataParallelGrads in eager mode (tests/test_gpu_determinism.py)."""
    from tests.test_gpu_determinism import LocalDP
    dp = LocalDP()
    dp.mode = mode
    return dp


def worker(mode, out):
    import torch
    from rick_amd.models import Discriminator, Generator
    from rick_amd.synth import synth_latents, synth_reals, synth_state_dict, synth_tensor
    from rick_amd.train import RickTrainer, TrainConfig
    from tests.shapes import discriminator_shapes, generator_shapes
    size, B, dev = int(os.environ.get('DIAG_SIZE', '32')), 2, 'cuda:0'

    def build():
        g = Generator(size, 512, 8, channel_multiplier=2)
        d = Discriminator(size, channel_multiplier=2)
        g.load_state_dict(synth_state_dict(generator_shapes(size)), strict=False)
        d.load_state_dict(synth_state_dict(discriminator_shapes(size)), strict=False)
        return g.to(dev), d.to(dev)
    g, d = build()
    g_ema, d_ema = build()
    dp = _LocalDP(mode)
    tr = RickTrainer(TrainConfig(size=size, batch=B, warmup_iter=1), g, d, g_ema, d_ema, dp=dp)
    z = synth_latents(B, seed=100).to(dev)
    real = synth_reals(B, size=size, seed=200).to(dev)
    noises = [synth_tensor(f'dpnoise/{i}', tuple(getattr(g.noises, f'noise_{i}').shape)).to(dev) for i in range(g.num_layers)]
    pl_noise = synth_tensor('dp/pl0', (1, 3, size, size)).to(dev)
    snaps = {}

    def snap(tag):
        torch.cuda.synchronize()
        for nm, fp in (('g', tr.g_flat), ('d', tr.d_flat)):
            snaps[f'{tag}/{nm}/grad'] = fp.grad.cpu().numpy().copy()
            snaps[f'{tag}/{nm}/flat'] = fp.flat.cpu().numpy().copy()
    tr.d_step(real, [z], i=0, g_noise=noises); snap('0_d_warm')
    tr.r1_step(real, i=0); snap('1_r1_warm')
    tr.d_step(real, [z], i=1, g_noise=noises); snap('2_d')
    tr.g_step([z], g_noise=noises); snap('3_g')
    tr.plr_step([z[:1]], pl_noise=pl_noise, g_noise=noises); snap('4_plr')
    tr.d_step(real, [z], i=2, g_noise=noises); snap('5_d')
    tr.r1_step(real, i=16); snap('6_r1')
    np.savez(out, **snaps)
    print(mode, 'hooks fired', dp.fired, flush=True)


def compare(fa, fb, label):
    from rick_amd.models import Discriminator, Generator
    from rick_amd.train import FlatParams, d_optim_filter, g_optim_filter
    size = int(os.environ.get('DIAG_SIZE', '32'))
    layout = {'g': FlatParams(Generator(size, 512, 8).named_parameters(), g_optim_filter),
              'd': FlatParams(Discriminator(size).named_parameters(), d_optim_filter)}
    a, b = np.load(fa), np.load(fb)
    bad = 0
    for k in sorted(a.files):
        if np.array_equal(a[k], b[k]):
            continue
        bad += 1
        fp = layout[k.split('/')[1]]
        print(f'[{label}] {k}: DIFFERS')
        for n in fp.names:
            lo, hi = fp.segment(n)
            x, y = a[k][lo:hi], b[k][lo:hi]
            ne = int((x != y).sum())
            if ne:
                print(f'    {n:46s} {ne:9d} / {hi - lo:9d} differ, max |d| {np.abs(x - y).max():.3e} (max |v| {np.abs(y).max():.3e})')
        if bad >= 3:
            print('    ... (later snapshots inherit the difference)')
            break
    if not bad:
        print(f'[{label}] identical in all {len(a.files)} snapshots')
    return bad


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'worker':
        worker(sys.argv[2], sys.argv[3])
        sys.exit(0)
    outdir = os.path.join(ROOT, 'gpurun_out', 'diag_det')
    os.makedirs(outdir, exist_ok=True)
    modes = sys.argv[1:] or ['plain', 'sync', 'copy']
    files = {}
    for mode in modes:
        for rep in range(2):
            f = os.path.join(outdir, f'{mode}_{rep}.npz')
            r = subprocess.run([sys.executable, os.path.abspath(__file__), 'worker', mode, f], capture_output=True, text=True)
            print(r.stdout[-500:], r.stderr[-1500:] if r.returncode else '')
            files[(mode, rep)] = f
    for mode in modes:
        compare(files[(mode, 0)], files[(mode, 1)], f'{mode} run0 vs run1')
    for mode in modes[1:]:
        compare(files[(modes[0], 0)], files[(mode, 0)], f'{modes[0]} vs {mode}')
    for f in files.values():
        os.remove(f)
