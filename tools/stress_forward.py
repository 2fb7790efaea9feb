"""Which layer is not run-to-run deterministic?  G forward (no_grad) -> D forward on cat(fake, real), repeated REPS times on
the same inputs in one process; a forward hook on every sub-module compares its output with the first repetition's ON THE
DEVICE and the first module whose output differs is reported (with the count and size of the difference).

    python tools/stress_forward.py [--reps 300] [--size 32] [--procs 1] [--grad]"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(args):
    import torch
    from rick_amd.synth import synth_latents, synth_reals, synth_tensor
    from rick_amd.models import Discriminator, Generator
    from rick_amd.synth import synth_state_dict
    from tests.shapes import discriminator_shapes, generator_shapes
    size, B, dev = args.size, args.batch, 'cuda:0'
    g = Generator(size, 512, 8, channel_multiplier=2)
    d = Discriminator(size, channel_multiplier=2)
    g.load_state_dict(synth_state_dict(generator_shapes(size)), strict=False)
    d.load_state_dict(synth_state_dict(discriminator_shapes(size)), strict=False)
    g, d = g.to(dev), d.to(dev)
    z = synth_latents(B, seed=100).to(dev)
    real = synth_reals(B, size=size, seed=200).to(dev)
    noises = [synth_tensor(f'dpnoise/{i}', tuple(getattr(g.noises, f'noise_{i}').shape)).to(dev) for i in range(g.num_layers)]
    ref, state = {}, {'rep': 0, 'first': None}
    counts = {}

    def flat(out):
        if torch.is_tensor(out):
            return [out]
        if isinstance(out, (tuple, list)):
            return [t for o in out for t in flat(o)]
        return []

    def make(name):
        def hook(mod, inp, out):
            for j, t in enumerate(flat(out)):
                key = f'{name}#{j}'
                if state['rep'] == 0:
                    ref[key] = t.detach().clone()
                elif state['first'] is None and not torch.equal(t.detach(), ref[key]):
                    dd = (t.detach() - ref[key]).abs()
                    state['first'] = (key, int((t.detach() != ref[key]).sum()), t.numel(), float(dd.max()), float(ref[key].abs().max()))
        return hook
    for prefix, net in (('g', g), ('d', d)):
        for name, mod in net.named_modules():
            if name:
                mod.register_forward_hook(make(f'{prefix}.{name}'))
    # the two banks are not modules: wrap their calls
    from rick_amd.op import modconv as _mc

    def wrap(cls, name):
        orig = cls.__call__

        def call(self, *a, **k):
            out = orig(self, *a, **k)
            if out is not None:
                make(name)(None, None, list(out))
            return out
        cls.__call__ = call
    wrap(_mc.ModulationBank, 'bank.modulation')
    wrap(_mc.DemodBank, 'bank.demod')
    bad = 0
    for rep in range(args.reps):
        state['rep'], state['first'] = rep, None
        if args.grad:
            for p in list(g.parameters()) + list(d.parameters()):
                p.grad = None
            fake, _ = g([z], noise=noises)
            pred, _ = d(torch.cat([fake, real], 0), calls=2)
            pred.square().sum().backward()
            gsum = {n: p.grad.clone() for n, p in list(g.named_parameters()) + list(d.named_parameters()) if p.grad is not None}
            if rep == 0:
                gref = gsum
            elif state['first'] is None:
                for n in gsum:
                    if not torch.equal(gsum[n], gref[n]):
                        state['first'] = ('grad of ' + n, int((gsum[n] != gref[n]).sum()), gsum[n].numel(),
                                          float((gsum[n] - gref[n]).abs().max()), float(gref[n].abs().max()))
                        break
        else:
            with torch.no_grad():
                fake, _ = g([z], noise=noises)
                d(torch.cat([fake, real], 0), calls=2)
        if state['first'] is not None:
            bad += 1
            counts[state['first'][0]] = counts.get(state['first'][0], 0) + 1
            if bad <= 8:
                k, ne, n, md, mv = state['first']
                print(f'[pid {os.getpid()}] rep {rep}: first difference at {k}: {ne}/{n} elements, max |d| {md:.3e} (max |v| {mv:.3e})', flush=True)
    print(f'[pid {os.getpid()}] {bad} of {args.reps} repetitions differed; first-difference sites: {counts}', flush=True)
    return bad


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=300)
    ap.add_argument('--procs', type=int, default=1)
    ap.add_argument('--size', type=int, default=32)
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--grad', action='store_true')
    ap.add_argument('--worker', action='store_true')
    args = ap.parse_args()
    if args.worker:
        sys.exit(1 if worker(args) else 0)
    cmd = [sys.executable, os.path.abspath(__file__), '--worker', '--reps', str(args.reps), '--size', str(args.size), '--batch', str(args.batch)]
    cmd += ['--grad'] if args.grad else []
    procs = [subprocess.Popen(cmd) for _ in range(args.procs)]       # (this parent never touches the GPU)
    rcs = [p.wait() for p in procs]
    print('exit codes', rcs)
