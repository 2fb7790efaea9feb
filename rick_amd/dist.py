"""Process-per-GPU data parallelism for the RICK loop over RCCL / xGMI.

The reference's only live parallelism is single-process ``nn.DataParallel`` on D
(train_dynamic_update_prune.py:941-944): per-call parameter broadcast, scatter, gather of 14
feature maps, reduce-add of gradients.  Here every rank owns a full replica and a micro-batch;
the only data-path exchange is the gradient average before each optimiser step:

  * gradients of one network live in ONE flat buffer (rick_amd.train.FlatParams), cut into
    contiguous buckets (default 32 MiB: large enough that a ring over 7 xGMI links runs at link
    rate, small enough that the first bucket leaves while backward is still producing the rest);
  * a post-accumulate-grad hook per parameter counts down its bucket; when a bucket is complete
    its ``all_reduce`` is issued asynchronously (torch.distributed 'nccl' == RCCL on ROCm; it
    runs on the backend's own stream), so communication overlaps the rest of backward;
  * ``all_reduce(flat)`` (called right before the optimiser step) issues whatever is left and
    waits.  Semantics: mean over ranks == the reference's loss ``.mean()`` over the global batch
    (RCCL's ncclAvg: no separate scaling pass); minibatch-stddev stays per rank, as under
    DataParallel's per-device chunks;
  * graph mode (RickTrainer.enable_graphs): ``launch(flat)`` issues every bucket right behind the
    replayed forward/backward graph and returns; the optimiser graph of that step is DEFERRED
    (``wait(flat)`` + replay) until the next piece of work that needs the updated parameters —
    the generator forward of the following G step runs while the D gradients are on the wire.

Fisher sweep: samples are sharded over ranks, grad^2 is reduced per filter locally (linear), and
only the per-filter vectors (~20 KB) are summed across ranks (``all_reduce_vectors``).
Works with the 'gloo' backend on CPU tensors too (used by the world_size-2 tests).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, local_rank, world)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = os.environ.get('RICK_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class DataParallelGrads:
    def __init__(self, bucket_bytes=32 << 20, group=None, force=False):
        """force: run the collectives even with a single rank (the RCCL code path on a one-GPU box: tests)."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.bucket_elems = max(1, bucket_bytes // 4)
        self._state = {}       # id(flat) -> dict(buckets, pending, works, hooks)
        self.hooks_enabled = True   # False: no launches from autograd hooks (backward replayed from a hipGraph)
        self.coalesce = True        # launch(): one collective when no bucket has left yet (False: always per bucket)

    # ------------------------------------------------------------------ bucket bookkeeping
    def attach(self, *flats):
        for flat in flats:
            buckets, cur_lo, cur_members = [], None, []
            first = list(getattr(flat, 'opt_idx', range(len(flat.params))))[0]
            # parameters are laid out in forward order; gradients arrive roughly in reverse, so
            # buckets are cut from the END of the buffer towards the front
            for i in reversed(getattr(flat, 'opt_idx', range(len(flat.params)))):
                lo, hi = int(flat.offsets[i]), int(flat.offsets[i + 1])
                if cur_lo is None:
                    cur_hi = hi
                cur_lo = lo
                cur_members.append(i)
                if cur_hi - cur_lo >= self.bucket_elems or i == first:
                    buckets.append({'lo': cur_lo, 'hi': cur_hi, 'members': list(cur_members)})
                    cur_lo, cur_members = None, []
            st = {'buckets': buckets, 'owner': {}, 'pending': [], 'works': [], 'launched': [], 'flat': flat}
            for b, bk in enumerate(buckets):
                for i in bk['members']:
                    st['owner'][i] = b
            self._state[id(flat)] = st
            self._arm(st)
            if self.active:
                for i in st['owner']:
                    flat.params[i].register_post_accumulate_grad_hook(self._make_hook(st, i))

    def _arm(self, st):
        st['pending'] = [sum(1 for i in bk['members'] if st['flat'].params[i].requires_grad) for bk in st['buckets']]
        st['launched'] = [False] * len(st['buckets'])
        st['works'] = []

    def _launch(self, st, b, whole=False):
        """bucket b — or (`whole`) the contiguous range of ALL buckets as one collective"""
        bk = st['buckets'][b]
        view = st['flat'].grad[bk['lo']:bk['hi']]
        if whole:
            view = st['flat'].grad[min(k['lo'] for k in st['buckets']):max(k['hi'] for k in st['buckets'])]
        if view.is_cuda and dist.get_backend(self.group) == 'gloo':
            # functional-test path only (two ranks sharing one GPU cannot use RCCL): stage through the host
            host = view.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            view.copy_(host)
            st['works'].append((None, view, 1.0 / self.world))
        elif dist.get_backend(self.group) == 'nccl':
            # RCCL: the mean itself (ncclAvg), asynchronous on the backend's stream behind the current stream's work
            st['works'].append((dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=True), view, None))
        else:
            st['works'].append((dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True), view, 1.0 / self.world))
        if whole:
            st['launched'] = [True] * len(st['buckets'])
        st['launched'][b] = True

    def _make_hook(self, st, i):
        def hook(_param):
            if not self.hooks_enabled:
                return
            b = st['owner'][i]
            st['pending'][b] -= 1
            if st['pending'][b] == 0 and not st['launched'][b]:
                self._launch(st, b)
        return hook

    # --------------------------------------------------------------------------- public
    def prepare(self, flat):
        """Re-count the gradients each bucket waits for from the parameters' CURRENT requires_grad
        flags.  Call after changing flags (warm-up gating, frozen D in the G step) and before
        backward: a bucket armed with stale flags could leave before all of its gradients exist."""
        if self.active:
            self._arm(self._state[id(flat)])

    def launch(self, flat):
        """Issue the all-reduce of every bucket that has not left yet and return without waiting."""
        if not self.active:
            return
        st = self._state[id(flat)]
        if self.coalesce and len(st['buckets']) > 1 and not any(st['launched']):
            # nothing has left yet (graph mode: the whole backward was ONE replayed graph, so bucketing buys no overlap): one
            # collective over the contiguous gradient range instead of one per bucket — on one GPU each collective costs
            # ~0.15 ms of stream hand-overs (bench.py dp1_forced: 0.955 -> see DESIGN.md section 6), on N GPUs one long ring
            # pays its latency terms once.  Same elementwise mean.
            self._launch(st, 0, whole=True)
            return
        for b in range(len(st['buckets'])):
            if not st['launched'][b]:
                self._launch(st, b)

    def wait(self, flat):
        """Make the current stream wait for the launched buckets (and scale where the backend summed)."""
        if not self.active:
            return
        st = self._state[id(flat)]
        for work, view, scale in st['works']:
            if work is not None:
                work.wait()
            if scale is not None:
                view.mul_(scale)
        self._arm(st)

    def all_reduce(self, flat):
        """Average flat.grad over ranks (finishes the buckets the hooks already started)."""
        self.launch(flat)
        self.wait(flat)

    def all_reduce_vectors(self, vectors):
        """Sum small vectors (per-filter Fisher) over ranks in one collective."""
        if not self.active:
            return
        flat = torch.cat([v.reshape(-1) for v in vectors])
        if flat.is_cuda and dist.get_backend(self.group) == 'gloo':
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            flat = host.to(flat.device)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for v in vectors:
            n = v.numel()
            v.copy_(flat[off:off + n].view_as(v))
            off += n

    def broadcast_params(self, modules, src=0):
        """Make replicas identical at start-up (one-off)."""
        if not self.active:
            return
        for m in modules:
            for t in list(m.parameters()) + list(m.buffers()):
                if t.is_cuda and dist.get_backend(self.group) == 'gloo':
                    host = t.data.cpu()
                    dist.broadcast(host, src=src, group=self.group)
                    t.data.copy_(host)
                else:
                    dist.broadcast(t.data, src=src, group=self.group)
