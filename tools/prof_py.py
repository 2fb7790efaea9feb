import sys, os, cProfile, pstats, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rick_amd.models import Discriminator, Generator
from rick_amd.train import mixing_noise
torch.manual_seed(1)
dev = 'cuda'
g = Generator(256, 512, 8).to(dev)
z = [torch.randn(4, 512, device=dev)]
with torch.no_grad():
    for _ in range(3): g(z)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10): g(z)
    pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
