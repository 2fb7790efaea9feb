import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.op import fused_act
from tools.bench_conv_util import timeit
for B in (4, 8):
    for c, r in [(512, 64), (256, 128), (128, 256)]:
        g = torch.randn(B, c, r, r, device='cuda').contiguous(memory_format=torch.channels_last)
        y = torch.randn_like(g)
        noise = torch.randn(B, 1, r, r, device='cuda')
        t = timeit(lambda: fused_act._ActAdjoint.apply(g, y, noise, 0.2, 2 ** 0.5, True, True), reps=30)
        print(f'cap={os.environ.get("RICK_BAB_CAP")} B={B} C={c} @{r}: {t*1e6:6.1f} us {3*g.numel()*4/t/1e12:5.2f} TB/s')
