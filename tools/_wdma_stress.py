import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.op import conv as cv
bad = 0
for (ci, co, r, B) in [(128, 256, 256, 4), (256, 512, 128, 8), (512, 512, 64, 8), (512, 512, 64, 4), (256, 512, 128, 2)]:
    torch.manual_seed(ci + r)
    x = torch.randn(B, ci, r + 1, r + 1, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 3, 3, device='cuda')
    wp = cv._pack(w, 1.0)
    os.environ['RICK_WDMA2'] = '0'
    ref = cv._conv_launch(x, wp, co, 3, 3, 2, 0)
    os.environ['RICK_WDMA2'] = '1'
    for it in range(40):
        y = cv._conv_launch(x, wp, co, 3, 3, 2, 0)
        if not torch.equal(y, ref):
            bad += 1
            print('MISMATCH', ci, co, r, B, it, float((y - ref).abs().max()))
    print('shape', ci, co, r, B, 'ok')
print('bad', bad)
