import sys, os, math
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd import op
import rick_amd.op.conv as cv
from rick_amd.models import ResBlock
from rick_amd.synth import synth_tensor, synth_state_dict
N, C = 4, 512
blk = ResBlock(C, C)
sd = synth_state_dict({kk: v.shape for kk, v in blk.state_dict().items()})
blk.load_state_dict(sd, strict=False); blk = blk.cuda()
x = synth_tensor('rb/x4', (N, C, 8, 8))
sc3 = 1 / math.sqrt(C * 9)
g1 = synth_tensor('rb/g1', (N, C, 8, 8)).cuda()
cap = {}
orig = cv._convT_launch
def spy(xin, wp, O, kh, kw, s, p, out_hw, **kw_):
    y = orig(xin, wp, O, kh, kw, s, p, out_hw, **kw_)
    cap['x'] = xin.detach().clone(); cap['xptr'] = xin.data_ptr(); cap['y'] = y.detach().clone(); cap['wp'] = wp
    cap['xobj'] = xin
    y2 = orig(xin, wp, O, kh, kw, s, p, out_hw, **kw_)
    cap['y2'] = y2.detach().clone()
    return y
cv._convT_launch = spy
xd = x.cuda().requires_grad_(True)
(ga,) = torch.autograd.grad(blk.conv1(xd), xd, g1)
cv._convT_launch = orig
torch.cuda.synchronize()
w = blk.conv1[0].weight.detach().double().cpu()
gin = cap['x'].double().cpu()
ref = F.conv_transpose2d(gin, w * sc3, stride=1, padding=1)
def perimg(a): return ['%.1e' % float((a[i].double().cpu() - ref[i]).abs().max()) for i in range(N)]
print('captured launch y vs ref on captured input:', perimg(cap['y']))
print('second identical launch y2:', perimg(cap['y2']))
print('xin strides', cap['xobj'].stride(), 'is_contig_cl', cap['xobj'].is_contiguous(memory_format=torch.channels_last), 'ptr%16', cap['xptr'] % 16, 'ptr%256', cap['xptr'] % 256)
print('x nan/inf', bool(torch.isnan(cap['x']).any()), bool(torch.isinf(cap['x']).any()), 'absmax', float(cap['x'].abs().max()), 'absmin nonzero', float(cap['x'][cap['x'] != 0].abs().min()))
# fresh copy of the same input at a new address
y3 = orig(cap['x'].clone(memory_format=torch.channels_last), cap['wp'], C, 3, 3, 1, 1, (8, 8))
print('launch on a fresh copy:', perimg(y3))
