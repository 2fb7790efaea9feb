"""Unhinted (create_graph route chosen inside the fused ops' backward) vs op.second_order() hinted graphs: first- and
second-order gradients of single layers and of the whole generator's path-length term."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd import op  # noqa: E402
from rick_amd.models import Generator, StyledConv, ToRGB  # noqa: E402

DEV = 'cuda'
torch.manual_seed(0)


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def layer_case(name, mod, x, style, extra=()):
    mod = mod.to(DEV)
    for p in mod.parameters():
        torch.nn.init.normal_(p, std=0.5) if p.ndim <= 1 else None
    res = {}
    for mode in ('hint', 'auto'):
        xs = x.clone().requires_grad_(True)
        st = style.clone().requires_grad_(True)
        with op.second_order(mode == 'hint'):
            y = mod(xs, st, *extra)
            gy = torch.randn(y.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
            gs, gx = torch.autograd.grad((y * gy).sum(), (st, xs), create_graph=True)
            pl = gs.pow(2).sum() + gx.pow(2).sum()
            gg = torch.autograd.grad(pl, [st, xs] + [p for p in mod.parameters()], allow_unused=True)
        res[mode] = (y.detach(), gs.detach(), gx.detach(), [g.detach() if g is not None else None for g in gg])
    h, a = res['hint'], res['auto']
    print(f'{name:28s} y {rel(a[0], h[0]):.1e} gs {rel(a[1], h[1]):.1e} gx {rel(a[2], h[2]):.1e} | second order: ' +
          ' '.join(f'{n}:{"none" if (ga is None or gh is None) else f"{rel(ga, gh):.1e}"}'
                   for n, ga, gh in zip(['st', 'x'] + [k for k, _ in mod.named_parameters()], a[3], h[3])))


B, SD = 2, 512
style = torch.randn(B, SD, device=DEV)
layer_case('StyledConv plain 64->64 8x8', StyledConv(64, 64, 3, SD), torch.randn(B, 64, 8, 8, device=DEV), style,
           (torch.randn(B, 1, 8, 8, device=DEV),))
layer_case('StyledConv up 64->64 8->16', StyledConv(64, 64, 3, SD, upsample=True), torch.randn(B, 64, 8, 8, device=DEV), style,
           (torch.randn(B, 1, 16, 16, device=DEV),))
layer_case('ToRGB 64 (no skip)', ToRGB(64, SD, upsample=False), torch.randn(B, 64, 8, 8, device=DEV), style)
skip = torch.randn(B, 3, 4, 4, device=DEV)
layer_case('ToRGB 64 (skip)', ToRGB(64, SD), torch.randn(B, 64, 8, 8, device=DEV), style, (skip,))

g = Generator(32, 512, 8).to(DEV)
z = torch.randn(2, 512, device=DEV)
noise = torch.randn(2, 3, 32, 32, device=DEV)
out = {}
for mode in ('hint', 'auto'):
    with op.second_order(mode == 'hint'):
        img, lat = g([z], return_latents=True, randomize_noise=False)
        (gr,) = torch.autograd.grad((img * noise).sum(), lat, create_graph=True)
        pl = gr.pow(2).sum(2).mean(1).sqrt()
        gp = torch.autograd.grad(pl.pow(2).mean(), [p for _, p in g.named_parameters()], allow_unused=True)
    out[mode] = (img.detach(), gr.detach(), pl.detach(), gp)
print('generator: img', rel(out['auto'][0], out['hint'][0]), 'dlat', rel(out['auto'][1], out['hint'][1]), 'pl', rel(out['auto'][2], out['hint'][2]))
per = (out['auto'][1] - out['hint'][1]).abs().amax(dim=(0, 2)) / out['hint'][1].abs().amax()
print('  per latent row:', ' '.join(f'{float(v):.1e}' for v in per))
for (k, _), ga, gh in zip(g.named_parameters(), out['auto'][3], out['hint'][3]):
    if (ga is None) != (gh is None):
        print('  ', k, 'auto', ga is None, 'hint', gh is None)
    elif ga is not None and rel(ga, gh) > 1e-4:
        print(f'   {k:40s} {rel(ga, gh):.1e}')
