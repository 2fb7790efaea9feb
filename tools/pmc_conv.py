"""One conv shape, a few launches — target for `rocprofv3 --pmc ...` (MFMA busy, HBM bytes)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.op import conv as cv
B, ci, co, r = 4, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(B, ci, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
w = torch.randn(co, ci, 3, 3, device='cuda')
gy = torch.randn(B, co, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
wp = cv._pack(w, 1.0)
for _ in range(6):
    cv._conv_launch(x, wp, co, 3, 3, 1, 1)
for _ in range(6):
    cv._wgrad_launch(gy, x, 3, 3, 1, 1)
torch.cuda.synchronize()
