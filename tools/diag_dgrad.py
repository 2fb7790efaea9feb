import sys, os, math
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd import op
from rick_amd.models import ResBlock
from rick_amd.synth import synth_tensor, synth_state_dict
from oracle.ops_ref import fused_leaky_relu_ref
N, C = 4, 512
blk = ResBlock(C, C)
sd = synth_state_dict({kk: v.shape for kk, v in blk.state_dict().items()})
blk.load_state_dict(sd, strict=False); blk = blk.cuda()
x = synth_tensor('rb/x4', (N, C, 8, 8))
sc3 = 1 / math.sqrt(C * 9)
xr = x.double().requires_grad_(True)
t1p = F.conv2d(xr, sd['conv1.0.weight'].double() * sc3, padding=1)
t1 = fused_leaky_relu_ref(t1p, sd['conv1.1.bias'].double())
g1 = synth_tensor('rb/g1', t1.shape)
(gp_r,) = torch.autograd.grad(t1, t1p, g1.double(), retain_graph=True)
(ga_r,) = torch.autograd.grad(t1, xr, g1.double())
res = []
for rep in range(3):
    xd = x.cuda().requires_grad_(True)
    (ga,) = torch.autograd.grad(blk.conv1(xd), xd, g1.cuda())
    res.append(ga.detach().double().cpu())
print('repeatable:', torch.equal(res[0], res[1]), torch.equal(res[1], res[2]))
err = (res[0] - ga_r).abs()
print('max err', float(err.max()), 'ref max', float(ga_r.abs().max()))
for n in range(N):
    e = err[n]
    print(n, 'max', float(e.max()), 'n_bad(>1e-4*max)', int((e > 1e-4 * ga_r.abs().max()).sum()), 'of', e.numel())
e = err[2]
bad = (e > 1e-4 * ga_r.abs().max()).nonzero()
print('bad idx sample (c,y,x):', bad[:20].tolist())
print('bad channels unique count', len(set(bad[:, 0].tolist())), 'ys', sorted(set(bad[:, 1].tolist())), 'xs', sorted(set(bad[:, 2].tolist())))
# direct op path with the same g (pre-act gradient from the oracle): isolates dgrad from act-bwd
w = blk.conv1[0].weight
y = op.conv2d(x.cuda().requires_grad_(True), w, 1, 1, wscale=sc3)
xd = x.cuda().requires_grad_(True)
(gd,) = torch.autograd.grad(op.conv2d(xd, w, 1, 1, wscale=sc3), xd, gp_r.float().cuda())
e2 = (gd.double().cpu() - ga_r).abs()
print('dgrad alone with oracle pre-act grad: per-image max err', [float(e2[i].max()) for i in range(N)])
