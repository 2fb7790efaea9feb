import sys, os, math
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd import op
from rick_amd.models import ResBlock
from rick_amd.synth import synth_tensor, synth_state_dict
N, C = 4, 512
blk = ResBlock(C, C)
sd = synth_state_dict({kk: v.shape for kk, v in blk.state_dict().items()})
blk.load_state_dict(sd, strict=False); blk = blk.cuda()
x = synth_tensor('rb/x4', (N, C, 8, 8))
sc3 = 1 / math.sqrt(C * 9)
w = blk.conv1[0].weight; b = blk.conv1[1].bias
g1 = synth_tensor('rb/g1', (N, C, 8, 8)).cuda()
t1p = op.conv2d(x.cuda(), w, 1, 1, wscale=sc3).detach().requires_grad_(True)
t1 = op.fused_leaky_relu(t1p, b)
(gp,) = torch.autograd.grad(t1, t1p, g1)
# reference on the device's own pre-activation values (CPU fp64)
tp = t1p.detach().double().cpu(); bb = b.detach().double().cpu()
pre = tp + bb.view(1, -1, 1, 1)
y_ref = torch.where(pre > 0, pre, pre * 0.2) * math.sqrt(2)
gp_ref = g1.double().cpu() * torch.where(y_ref > 0, 1.0, 0.2) * math.sqrt(2)
print('t1 fwd err', float((t1.double().cpu() - y_ref).abs().max()))
e = (gp.double().cpu() - gp_ref).abs()
print('gp max err', float(e.max()), 'count > 1e-6:', int((e > 1e-6).sum()))
bad = (e > 1e-6).nonzero()
print('bad sample', bad[:10].tolist())
if len(bad):
    n, c, yy, xx = bad[0].tolist()
    print('at bad[0]: pre', float(pre[n, c, yy, xx]), 'y_dev', float(t1[n, c, yy, xx]), 'y_ref', float(y_ref[n, c, yy, xx]),
          'g', float(g1[n, c, yy, xx]), 'gp_dev', float(gp[n, c, yy, xx]), 'gp_ref', float(gp_ref[n, c, yy, xx]))
    print('bad pixels', sorted(set((r[0], r[2], r[3]) for r in bad.tolist()))[:10], 'nch', len(set(r[1] for r in bad.tolist())))
print('strides t1p', t1p.stride(), 't1', t1.stride(), 'gp', gp.stride(), 'g1', g1.stride())
