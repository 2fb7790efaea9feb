"""Narrowing down the run-to-run differences of the modulation bank's output when two processes share the GPU
(tools/stress_forward.py found them at `bank.modulation`): stages of growing context around the same launch.

    python tools/stress_bank.py [--reps 1500] [--procs 2]"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(args):
    import torch
    from rick_amd import op
    from rick_amd.models import Discriminator, Generator
    from rick_amd.op import split as sp
    from rick_amd.synth import synth_latents, synth_reals, synth_state_dict, synth_tensor
    from tests.shapes import discriminator_shapes, generator_shapes
    size, B, dev = 32, 2, 'cuda:0'
    g = Generator(size, 512, 8, channel_multiplier=2)
    d = Discriminator(size, channel_multiplier=2)
    g.load_state_dict(synth_state_dict(generator_shapes(size)), strict=False)
    d.load_state_dict(synth_state_dict(discriminator_shapes(size)), strict=False)
    g, d = g.to(dev), d.to(dev)
    z = synth_latents(B, seed=100).to(dev)
    real = synth_reals(B, size=size, seed=200).to(dev)
    noises = [synth_tensor(f'dpnoise/{i}', tuple(getattr(g.noises, f'noise_{i}').shape)).to(dev) for i in range(g.num_layers)]
    bank = g._modulation_bank()
    with torch.no_grad():
        lat_fixed = g.style(z).unsqueeze(1).repeat(1, g.n_latent, 1).contiguous()

    def only_bank():
        return torch.cat([t.reshape(-1) for t in bank(lat_fixed)])

    def bank_after_empty_churn():
        junk = [torch.empty(1 << 14, device=dev).fill_(float(i)) for i in range(8)]      # allocator traffic + other writers before
        out = torch.cat([t.reshape(-1) for t in bank(lat_fixed)])
        del junk
        return out

    def bank_with_arena():
        words = [sp.new_amax(dev) for _ in range(6)]
        x = torch.randn(2, 64, 8, 8, device=dev)
        for w in words:
            sp.amax(x, w)
        return torch.cat([t.reshape(-1) for t in bank(lat_fixed)])

    def style_then_bank():
        lat = g.style(z).unsqueeze(1).repeat(1, g.n_latent, 1)
        return torch.cat([t.reshape(-1) for t in bank(lat)])

    def bank_then_demod():
        sb = bank(lat_fixed)
        dl = g._demod_bank()(sb)
        return torch.cat([t.reshape(-1) for t in sb] + [t.reshape(-1) for t in dl])

    def g_forward_bank_out(sync_before=False, sync_after=False, use_fixed=False):
        box = {}
        orig = type(bank).__call__

        def call(self, *a, **k):
            if sync_before:
                torch.cuda.synchronize()
            out = orig(self, lat_fixed) if use_fixed else orig(self, *a, **k)
            if sync_after:
                torch.cuda.synchronize()
            box['s'] = torch.cat([t.reshape(-1) for t in out]).clone()
            return out
        type(bank).__call__ = call
        try:
            g([z], noise=noises)
        finally:
            type(bank).__call__ = orig
        return box['s']

    def g_then_d_bank_out():
        s = g_forward_bank_out()
        fake, _ = g([z], noise=noises)
        d(torch.cat([fake, real], 0), calls=2)
        return s

    import functools
    stages = [('G forward, sync BEFORE the bank launch', functools.partial(g_forward_bank_out, True, False)),
              ('G forward, sync AFTER the bank launch', functools.partial(g_forward_bank_out, False, True)),
              ('G forward, bank on a FIXED latent', functools.partial(g_forward_bank_out, False, False, True)),
              ('bank only', only_bank), ('bank after allocator churn', bank_after_empty_churn), ('bank after amax launches', bank_with_arena),
              ('mapping + repeat + bank', style_then_bank), ('bank + demod bank', bank_then_demod),
              ('G forward (bank output)', g_forward_bank_out), ('G + D forward (bank output of G)', g_then_d_bank_out)]
    total = 0
    with torch.no_grad():
        for name, fn in stages:
            ref = fn().clone()
            bad = 0
            for _ in range(args.reps):
                bad += int(not torch.equal(fn(), ref))
            torch.cuda.synchronize()
            total += bad
            print(f'[pid {os.getpid()}] {name:36s} {bad:4d} / {args.reps} differ', flush=True)
    return total


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=1500)
    ap.add_argument('--procs', type=int, default=2)
    ap.add_argument('--worker', action='store_true')
    args = ap.parse_args()
    if args.worker:
        sys.exit(1 if worker(args) else 0)
    cmd = [sys.executable, os.path.abspath(__file__), '--worker', '--reps', str(args.reps)]
    procs = [subprocess.Popen(cmd) for _ in range(args.procs)]       # (this parent never touches the GPU)
    print('exit codes', [p.wait() for p in procs])
