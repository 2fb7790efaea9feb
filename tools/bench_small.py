"""GPU durations of the small-layer conv launches (run under rocprofv3 --kernel-trace --stats): 512->512 3x3 at 4^2 / 8^2 / 16^2."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.op import conv as cv
B = int(os.environ.get('B', 4))
for r in (4, 8, 16):
    x = torch.randn(B, 512, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(512, 512, 3, 3, device='cuda')
    wp = cv._pack(w, 1.0)
    for _ in range(int(os.environ.get('REPS', 20)) * (1 if r != 16 else 1)):
        cv._conv_launch(x, wp, 512, 3, 3, 1, 1)
    torch.cuda.synchronize()
