import sys, os, math
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd import op
from rick_amd.models import ResBlock
from rick_amd.synth import synth_tensor, synth_state_dict
from oracle.ops_ref import upfirdn2d_ref, fused_leaky_relu_ref, make_blur_kernel
def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max())
k = make_blur_kernel([1, 3, 3, 1]).double()
for N in (2, 4):
    C = 512
    blk = ResBlock(C, C)
    sd = synth_state_dict({kk: v.shape for kk, v in blk.state_dict().items()})
    blk.load_state_dict(sd, strict=False); blk = blk.cuda()
    x = synth_tensor(f'rb/x{N}', (N, C, 8, 8))
    xr = x.double().requires_grad_(True)
    sc3, sc1 = 1 / math.sqrt(C * 9), 1 / math.sqrt(C)
    t1p = F.conv2d(xr, sd['conv1.0.weight'].double() * sc3, padding=1); t1p.retain_grad()
    t1 = fused_leaky_relu_ref(t1p, sd['conv1.1.bias'].double()); t1.retain_grad()
    b2 = upfirdn2d_ref(t1, k, pad=(2, 2)); b2.retain_grad()
    t2p = F.conv2d(b2, sd['conv2.1.weight'].double() * sc3, stride=2); t2p.retain_grad()
    t2 = fused_leaky_relu_ref(t2p, sd['conv2.2.bias'].double()); t2.retain_grad()
    sb = upfirdn2d_ref(xr, k, pad=(1, 1)); sb.retain_grad()
    sk = F.conv2d(sb, sd['skip.1.weight'].double() * sc1, stride=2); sk.retain_grad()
    yr = (t2 + sk) / math.sqrt(2)
    gy = synth_tensor(f'rb/gy{N}', yr.shape)
    yr.backward(gy.double())
    # device, with hooks on intermediates
    xd = x.cuda().requires_grad_(True)
    grads = {}
    def keep(name):
        def h(g): grads[name] = g
        return h
    t1d = blk.conv1(xd); t1d.register_hook(keep('t1'))
    b2d = blk.conv2[0](t1d); b2d.register_hook(keep('b2'))
    t2pd = blk.conv2[1](b2d); t2pd.register_hook(keep('t2p'))
    t2d = blk.conv2[2](t2pd); t2d.register_hook(keep('t2'))
    sbd = blk.skip[0](xd); sbd.register_hook(keep('sb'))
    skd = blk.skip[1](sbd); skd.register_hook(keep('sk'))
    yd = op.add_scale(t2d, skd, 1 / math.sqrt(2))
    yd.backward(gy.cuda())
    print(f'N={N}: fwd {rel(yd, yr):.1e} | grads: t2 {rel(grads["t2"], t2.grad):.1e} t2p {rel(grads["t2p"], t2p.grad):.1e} '
          f'b2 {rel(grads["b2"], b2.grad):.1e} t1 {rel(grads["t1"], t1.grad):.1e} sk {rel(grads["sk"], sk.grad):.1e} '
          f'sb {rel(grads["sb"], sb.grad):.1e} x {rel(xd.grad, xr.grad):.1e}')
    for n_, p_ in blk.named_parameters():
        ref = {'conv1.0.weight': None}

print('--- branch isolation, N=4')
N, C = 4, 512
blk = ResBlock(C, C)
sd = synth_state_dict({kk: v.shape for kk, v in blk.state_dict().items()})
blk.load_state_dict(sd, strict=False); blk = blk.cuda()
x = synth_tensor('rb/x4', (N, C, 8, 8))
sc3, sc1 = 1 / math.sqrt(C * 9), 1 / math.sqrt(C)
# main branch only: t1 = conv1(x); grad wrt x given g on t1
xr = x.double().requires_grad_(True)
t1 = fused_leaky_relu_ref(F.conv2d(xr, sd['conv1.0.weight'].double() * sc3, padding=1), sd['conv1.1.bias'].double())
g1 = synth_tensor('rb/g1', t1.shape)
(ga_r,) = torch.autograd.grad(t1, xr, g1.double())
xd = x.cuda().requires_grad_(True)
(ga,) = torch.autograd.grad(blk.conv1(xd), xd, g1.cuda())
print('main branch dx', rel(ga, ga_r), 'per-image', [rel(ga[i], ga_r[i]) for i in range(N)])
# skip blur only
xr = x.double().requires_grad_(True)
sb = upfirdn2d_ref(xr, k, pad=(1, 1)); g2 = synth_tensor('rb/g2', sb.shape)
(gb_r,) = torch.autograd.grad(sb, xr, g2.double())
xd = x.cuda().requires_grad_(True)
(gb,) = torch.autograd.grad(blk.skip[0](xd), xd, g2.cuda())
print('skip blur dx', rel(gb, gb_r), 'per-image', [rel(gb[i], gb_r[i]) for i in range(N)])
# both, accumulated by autograd
xd = x.cuda().requires_grad_(True)
(gab,) = torch.autograd.grad([blk.conv1(xd), blk.skip[0](xd)], xd, [g1.cuda(), g2.cuda()])
print('both (autograd accumulate)', rel(gab, ga_r + gb_r), ' manual sum', rel(ga + gb, ga_r + gb_r))
print('strides ga', ga.stride(), 'gb', gb.stride(), 'gab', gab.stride())
