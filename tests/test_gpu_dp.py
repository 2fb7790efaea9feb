"""Row (e) on one GPU: two ranks of the RICK loop share cuda:0 and exchange gradients through
DataParallelGrads (bucketed, launched from post-accumulate hooks; gloo with host staging, because
RCCL refuses two ranks on one device).  The result must be bit-identical to the same two ranks
using a trivial blocking all-reduce of the whole flat buffer, across the warm-up -> full-D stage
change, the frozen-D G step and the second-order R1 / path-length steps."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


class _BlockingDP:
    """Reference exchange: no hooks, no buckets — average the whole gradient buffer after backward."""

    def __init__(self):
        self.world = torch.distributed.get_world_size()

    def attach(self, *flats):
        pass

    def prepare(self, flat):
        pass

    def all_reduce(self, flat):
        host = flat.grad.cpu()
        torch.distributed.all_reduce(host)
        flat.grad.copy_(host)
        flat.grad.mul_(1.0 / self.world)


def _worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    from rick_amd.dist import DataParallelGrads, init_from_env
    from rick_amd.models import Discriminator, Generator
    from rick_amd.synth import synth_latents, synth_reals, synth_state_dict, synth_tensor
    from rick_amd.train import RickTrainer, TrainConfig
    from tests.shapes import discriminator_shapes, generator_shapes
    init_from_env('gloo')
    size, B, dev = 32, 2, 'cuda:0'

    def build():
        g = Generator(size, 512, 8, channel_multiplier=2)
        d = Discriminator(size, channel_multiplier=2)
        g.load_state_dict(synth_state_dict(generator_shapes(size)), strict=False)
        d.load_state_dict(synth_state_dict(discriminator_shapes(size)), strict=False)
        return g.to(dev), d.to(dev)
    g, d = build()
    g_ema, d_ema = build()
    dp = DataParallelGrads(bucket_bytes=256 * 1024) if mode == 'bucketed' else _BlockingDP()
    tr = RickTrainer(TrainConfig(size=size, batch=B, warmup_iter=1), g, d, g_ema, d_ema, dp=dp)
    if mode == 'bucketed':
        assert len(dp._state[id(tr.d_flat)]['buckets']) >= 3
    z = synth_latents(B, seed=100 + rank).to(dev)                 # each rank its own micro-batch
    real = synth_reals(B, size=size, seed=200 + rank).to(dev)
    noises = [synth_tensor(f'dpnoise/{i}', tuple(getattr(g.noises, f'noise_{i}').shape)).to(dev)
              for i in range(g.num_layers)]
    pl_noise = synth_tensor(f'dp/pl{rank}', (1, 3, size, size)).to(dev)
    tr.d_step(real, [z], i=0, g_noise=noises)                     # warm-up: only final_* of D carries gradients
    tr.r1_step(real, i=0)
    tr.d_step(real, [z], i=1, g_noise=noises)                     # stage change: every D bucket now fills
    tr.g_step([z], g_noise=noises)                                # D frozen
    tr.plr_step([z[:1]], pl_noise=pl_noise, g_noise=noises)
    tr.d_step(real, [z], i=2, g_noise=noises)
    torch.cuda.synchronize()
    q.put((rank, tr.g_flat.flat.detach().cpu().numpy(), tr.d_flat.flat.detach().cpu().numpy()))
    torch.distributed.destroy_process_group()


def _run(mode):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=600) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return outs


def test_two_rank_trainer_bucketed_equals_blocking_allreduce():
    a = _run('bucketed')
    b = _run('blocking')
    for net in (1, 2):
        assert np.array_equal(a[0][net], a[1][net])               # replicas stay identical
        assert np.isfinite(a[0][net]).all()
        assert np.array_equal(a[0][net], b[0][net])               # same update as the plain exchange
