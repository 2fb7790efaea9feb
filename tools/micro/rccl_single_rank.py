"""Device time of a single-rank RCCL all-reduce (AVG) at the sizes of the gradient exchange (95 MB generator, 115 MB
discriminator, and one 32 MiB bucket): what `bench.py dp1_forced` pays for the collectives themselves on one GPU."""
import os, socket, sys, time
import torch
import torch.distributed as dist
os.dup2(2, 1)
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
for mb in (32 * 1.048576, 95, 115):
    n = int(mb * 1e6 / 4)
    t = torch.randn(n, device='cuda')
    for _ in range(3):
        dist.all_reduce(t, op=dist.ReduceOp.AVG)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dist.all_reduce(t, op=dist.ReduceOp.AVG)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f'{mb:7.1f} MB single-rank all_reduce(AVG): {us:8.1f} us  ({2 * n * 4 / us / 1e6:.2f} TB/s read+write)', file=sys.stderr)
dist.destroy_process_group()
