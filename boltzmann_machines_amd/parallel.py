"""Multi-GPU execution of the two paths that shard, one process per GPU.

* data-parallel CD-k: every rank runs the Gibbs chain and the raw outer products on its
  rows (`grad_step`), ONE all-reduce(sum) of the fused buffer [dW | sum(X-v) | sum(h0-hk)
  | sum(hk)] (RCCL over xGMI on the GPU box, enqueued on the engine's own HIP stream),
  then every rank applies the identical update with the GLOBAL batch size (`apply_step`),
  so the replicas stay bit-identical.  Sample bitmaps are a function of the GLOBAL row
  index (`set_row_offset`), i.e. independent of the number of ranks.
* AIS: the chains are independent; each rank runs a contiguous slice of the chains (the
  chain index is global in the RNG stream) and the per-chain log-weights are all-gathered
  once at the end (`log_Z()` returns them, reference dbm.py:922-923,939).

The collectives are injected (`allreduce_`, `allgather`), so the same code runs over
torch.distributed/RCCL on the GPU box and over gloo in the CPU tests.
"""
import os

import numpy as np


def shard(n, rank, world):
    """contiguous slice [start, stop) of n items for `rank` (remainder spread over the first ranks)"""
    q, r = divmod(n, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def dist_env():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


class DataParallelRBM(object):
    """Drives one rank's engine (RbmEngine, or any object with grad_step / apply_step /
    set_row_offset) through data-parallel CD-k updates."""

    def __init__(self, engine, rank, world, local_batch, allreduce_):
        self.engine, self.rank, self.world, self.local_batch = engine, rank, world, local_batch
        self.allreduce_ = allreduce_            # in-place sum over ranks of this rank's grad buffer
        engine.set_row_offset(rank * local_batch)

    def train_step(self, X_local, lr, momentum, k, **kw):
        self.engine.grad_step(X_local, self.local_batch, k, **kw)
        self.allreduce_()
        self.engine.apply_step(self.local_batch * self.world, lr, momentum)


def torch_allreduce_on_engine_stream(engine, device):
    """all-reduce of the engine's "grad" buffer by RCCL, enqueued on the engine's HIP stream
    (zero-copy: the buffer is wrapped through __cuda_array_interface__)."""
    import torch
    import torch.distributed as dist
    stream = torch.cuda.ExternalStream(engine.stream(), device=device)
    buf = torch.as_tensor(engine.device_view('grad'), device=device)

    def allreduce_():
        with torch.cuda.stream(stream):
            dist.all_reduce(buf)
    return allreduce_


def ais_sharded(run_ais, n_runs, rank, world, allgather):
    """run_ais(n_local, chain0) -> per-chain log Z estimates of chains [chain0, chain0+n_local);
    allgather(local ndarray, counts) -> concatenation over ranks.  Returns all n_runs values."""
    start, stop = shard(n_runs, rank, world)
    local = np.ascontiguousarray(run_ais(stop - start, start), dtype=np.float32)
    counts = [shard(n_runs, r, world)[1] - shard(n_runs, r, world)[0] for r in range(world)]
    return allgather(local, counts)


def torch_allgather(device=None):
    import torch
    import torch.distributed as dist

    def allgather(local, counts):
        m = max(counts)
        pad = np.zeros(m, dtype=np.float32)
        pad[:len(local)] = local
        t = torch.from_numpy(pad)
        if device is not None:
            t = t.to(device)
        outs = [torch.empty_like(t) for _ in counts]
        dist.all_gather(outs, t)
        return np.concatenate([o.cpu().numpy()[:c] for o, c in zip(outs, counts)])
    return allgather


class DataParallelDBM(object):
    """Data-parallel DBM update (BASELINE configs[3]): rank r owns rows [r*N_local, ...) of every
    minibatch (its own mean-field parameters) and particles [r*M_local, ...).  Per update:
    `grad_step` (mean-field with an all-reduce(max) of the residual per sweep, PCD, raw sums) ->
    ONE all-reduce(sum) of the fused buffer -> `apply_step` with the global N and M."""

    def __init__(self, engine, rank, world, allreduce_, allreduce_max=None):
        self.engine, self.rank, self.world = engine, rank, world
        self.allreduce_ = allreduce_
        engine.set_row_offset(rank * engine.N, rank * engine.M)
        if allreduce_max is not None and world > 1:
            engine.set_mf_allreduce(allreduce_max)

    def train_step(self, X_local, lr, momentum, k, **kw):
        n_mf = self.engine.grad_step(X_local, k, **kw)
        self.allreduce_()
        self.engine.apply_step(self.engine.N * self.world, self.engine.M * self.world, lr, momentum)
        return n_mf


def torch_allreduce_max():
    import torch
    import torch.distributed as dist

    def allreduce_max(x):
        t = torch.tensor([x], dtype=torch.float32)
        if dist.get_backend() == 'nccl':
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return allreduce_max
