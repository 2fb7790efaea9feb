"""Bit-reproducibility of individual ops while ANOTHER process keeps the same GPU busy with generator / discriminator passes
(two ranks share cuda:0 in tests/test_gpu_dp.py; in production RCCL kernels run next to the compute stream).  Every op is
evaluated REPS times on fixed inputs and compared with its first result on the device.

    python tools/stress_ops.py [--reps 2000] [--seconds 60]"""
import argparse
import importlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(size=32):
    from rick_amd.models import Discriminator, Generator
    from rick_amd.synth import synth_state_dict
    from tests.shapes import discriminator_shapes, generator_shapes
    g = Generator(size, 512, 8, channel_multiplier=2)
    d = Discriminator(size, channel_multiplier=2)
    g.load_state_dict(synth_state_dict(generator_shapes(size)), strict=False)
    d.load_state_dict(synth_state_dict(discriminator_shapes(size)), strict=False)
    return g.cuda(), d.cuda()


def aggressor(seconds):
    import torch
    g, d = build(64)
    z = torch.randn(4, 512, device='cuda')
    fake, _ = g([z])
    torch.cuda.synchronize()
    print('[aggressor] ready', flush=True)
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        fake, _ = g([z])
        pred, _ = d(fake)
        pred.square().sum().backward()
        n += 1
    torch.cuda.synchronize()
    print(f'[aggressor] {n} G + D forward / backward passes', flush=True)


def victim(reps, only=None, size=32):
    """-> number of evaluations that differed from the first one (`only`: substrings selecting ops)."""
    import torch
    from rick_amd import op
    from rick_amd.models import make_kernel
    L = importlib.import_module('rick_amd.op.linear')
    torch.manual_seed(0)
    dev = 'cuda'
    g, d = build(size)
    z = torch.randn(2, 512, device=dev)
    with torch.no_grad():
        lat = g.style(z).unsqueeze(1).repeat(1, g.n_latent, 1).contiguous()
    bank, dbank = g._modulation_bank(), g._demod_bank()
    x512 = torch.randn(2, 512, 16, 16, device=dev).contiguous(memory_format=torch.channels_last)
    x128 = torch.randn(2, 128, 64, 64, device=dev).contiguous(memory_format=torch.channels_last)
    w512 = torch.randn(512, 512, 3, 3, device=dev)
    w1 = torch.randn(1, 512, 512, 3, 3, device=dev)
    s = torch.rand(2, 512, device=dev) + 0.5
    k4 = make_kernel([1, 3, 3, 1]).to(dev)
    bias = torch.randn(512, device=dev)
    xl, wl, gl = torch.randn(8, 8192, device=dev), torch.randn(512, 8192, device=dev), torch.randn(8, 512, device=dev)
    w3 = torch.randn(3, 512, device=dev)
    img = torch.randn(4, 3, size, size, device=dev)

    def cat(ts):
        return torch.cat([t.reshape(-1) for t in ts])

    def modconv(up):
        from rick_amd.op import modconv as mc
        dd = mc.demod_coeff(w1[0], s, 1 / (512 * 9) ** 0.5, 1e-8)
        return mc.modulated_conv_fused(x512, w1[0], s, dd, 1 / (512 * 9) ** 0.5, up)

    def d_fwd_bwd():
        for p in d.parameters():
            p.grad = None
        out, _ = d(img)
        out.square().sum().backward()
        return cat([out] + [p.grad for p in d.parameters() if p.grad is not None])

    def g_fwd_bwd():
        for p in g.parameters():
            p.grad = None
        out, _ = g([lat.detach()], input_is_latent=True, randomize_noise=False)
        out.square().sum().backward()
        return cat([out] + [p.grad for p in g.parameters() if p.grad is not None])

    ops = [
        ('modulation bank fwd', lambda: cat(bank(lat))),
        ('demod bank fwd', lambda: cat(dbank(bank(lat)))),
        ('equal_linear (mapping layer)', lambda: op.equal_linear(z, g.style[1].weight, g.style[1].bias, g.style[1].scale, 0.01, True, True)),
        ('linear fwd 8192->512', lambda: L._p1(xl, wl, None, 0.01, 0.0)),
        ('linear dgrad', lambda: L._p2(gl, wl, 0.01)),
        ('linear wgrad', lambda: L._p3(gl, xl, 0.01)[0]),
        ('conv2d 3x3 512 @16', lambda: op.conv2d(x512, w512, 1, 1, wscale=0.01)),
        ('conv2d 3x3 stride 2', lambda: op.conv2d(x512, w512, 2, 0, wscale=0.01)),
        ('modulated conv', lambda: modconv(False)),
        ('modulated conv upsample', lambda: modconv(True)),
        ('upfirdn2d blur 128ch @64', lambda: op.upfirdn2d(x128, k4, pad=(2, 1))),
        ('upfirdn2d up2', lambda: op.upfirdn2d(x128, k4 * 4, up=2, pad=(2, 1))),
        ('upfirdn2d down2', lambda: op.upfirdn2d(x128, k4, down=2, pad=(1, 1))),
        ('fused_leaky_relu', lambda: op.fused_leaky_relu(x512, bias)),
        ('torgb', lambda: op.torgb(x512, w3, s, wscale=0.04)),
        ('minibatch stddev', lambda: op.minibatch_stddev(x512[:, :, :4, :4].repeat(2, 1, 1, 1).contiguous(), 4)),
        ('D forward + backward (all gradients)', d_fwd_bwd),
        ('G forward + backward (all gradients)', g_fwd_bwd),
    ]
    total = 0
    for name, fn in ops:
        if only is not None and not any(o in name for o in only):
            continue
        grad = 'backward' in name
        ctx = torch.enable_grad() if grad else torch.no_grad()
        with ctx:
            ref = fn().detach().clone()
            bad = 0
            n = max(50, reps // 10) if grad else reps
            for _ in range(n):
                bad += int(not torch.equal(fn().detach(), ref))
        torch.cuda.synchronize()
        total += bad
        print(f'[victim] {name:40s} {bad:5d} / {n} differ', flush=True)
    return total


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=2000)
    ap.add_argument('--seconds', type=float, default=90)
    ap.add_argument('--role', default='')
    ap.add_argument('--size', type=int, default=32)
    args = ap.parse_args()
    if args.role == 'aggressor':
        aggressor(args.seconds)
        sys.exit(0)
    if args.role == 'victim':
        sys.exit(1 if victim(args.reps, size=args.size) else 0)
    me = [sys.executable, os.path.abspath(__file__), '--reps', str(args.reps), '--seconds', str(args.seconds), '--size', str(args.size)]
    a = subprocess.Popen(me + ['--role', 'aggressor'])                 # (this parent never touches the GPU)
    time.sleep(20)
    v = subprocess.Popen(me + ['--role', 'victim'])
    rc = v.wait()
    a.wait()
    print('victim exit code', rc)
    sys.exit(rc)
