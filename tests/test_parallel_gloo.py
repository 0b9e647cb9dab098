"""world_size-2 tests of the sharded paths on CPU (gloo): the data-parallel CD-k algebra and
the AIS chain sharding of boltzmann_machines_amd/parallel.py.  The per-rank compute is the
CPU oracle twin (the HIP engine needs a GPU); what is under test is the N > 1 host logic:
row offsets, the fused all-reduce buffer, global-batch normalisation, chain-global RNG."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class TwinAsEngine(object):
    """adapts OracleRBM to the engine interface used by DataParallelRBM"""

    def __init__(self, twin):
        self.twin, self.raw = twin, None

    def set_row_offset(self, row0):
        self.twin.row0 = row0

    def grad_step(self, X, B, k):
        self.raw = self.twin.raw_grads(X, k)

    def apply_step(self, B_global, lr, mom):
        self.twin.apply(self.raw, float(B_global), lr, mom)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from boltzmann_machines_amd import parallel
    from oracle import oracle as orc
    dist.init_process_group('gloo', rank=rank, world_size=world)
    V, H, Bl, k = 24, 16, 6, 2
    kw = dict(sample_v_states=True, l2=1e-3, sparsity_cost=1e-2)
    W = (orc.normal(1, 2, 0, V * H) * np.float32(0.1)).reshape(V, H)
    Xg = (orc.uniform(1, 3, 0, world * Bl * V) < 0.3).astype(np.float32).reshape(world * Bl, V)
    twin = orc.OracleRBM(V, H, **kw)
    twin.p['W'][...] = W
    twin.set_seed(99)
    eng = TwinAsEngine(twin)

    def allreduce_():
        t = torch.from_numpy(eng.raw)
        dist.all_reduce(t)
    dp = parallel.DataParallelRBM(eng, rank, world, Bl, allreduce_)
    for step in range(3):
        dp.train_step(Xg[rank * Bl:(rank + 1) * Bl], 0.05, 0.5, k)
    # AIS sharding with the DBM oracle
    dbm = orc.OracleDBM(10, [8, 6], n_particles=4, batch_size=4)
    dbm.p['W'][...] = (orc.normal(5, 1, 0, 80) * np.float32(0.2)).reshape(10, 8)
    dbm.p['W_1'][...] = (orc.normal(5, 2, 0, 48) * np.float32(0.2)).reshape(8, 6)
    vals = parallel.ais_sharded(lambda n, c0: dbm.ais(12, n, 1, 777, chain0=c0), 7, rank, world,
                                parallel.torch_allgather())
    if rank == 0:
        np.savez(out, W=twin.p['W'], vb=twin.p['vb'], hb=twin.p['hb'], q=twin.p['q_means'], ais=vals,
                 hs_rank0=twin.work['hs'])
    else:
        np.savez(out + '.r1', hs_rank1=twin.work['hs'], W=twin.p['W'])
    dist.destroy_process_group()


def test_dp_and_ais_sharding_world2(tmp_path):
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    out = str(tmp_path / 'dp.npz')
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    got1 = np.load(out + '.r1.npz')
    # replicas identical after the update
    assert np.array_equal(got['W'], got1['W'])
    # single-process reference on the concatenated global batch
    V, H, Bl, k, world = 24, 16, 6, 2, 2
    kw = dict(sample_v_states=True, l2=1e-3, sparsity_cost=1e-2)
    ref = orc.OracleRBM(V, H, **kw)
    ref.p['W'][...] = (orc.normal(1, 2, 0, V * H) * np.float32(0.1)).reshape(V, H)
    ref.set_seed(99)
    Xg = (orc.uniform(1, 3, 0, world * Bl * V) < 0.3).astype(np.float32).reshape(world * Bl, V)
    for step in range(3):
        ref.train_step(Xg, 0.05, 0.5, k)
    # sums are blocked per rank, so parameters agree to fp32 round-off, and the sample bitmaps
    # (functions of the GLOBAL row index) are identical
    for n, key in (('W', 'W'), ('vb', 'vb'), ('hb', 'hb'), ('q_means', 'q')):
        np.testing.assert_allclose(got[key], ref.p[n], rtol=2e-5, atol=2e-7)
    assert np.array_equal(np.concatenate([got['hs_rank0'], got1['hs_rank1']]), ref.work['hs'])
    # AIS: sharded == unsharded, bit for bit
    dbm = orc.OracleDBM(10, [8, 6], n_particles=4, batch_size=4)
    dbm.p['W'][...] = (orc.normal(5, 1, 0, 80) * np.float32(0.2)).reshape(10, 8)
    dbm.p['W_1'][...] = (orc.normal(5, 2, 0, 48) * np.float32(0.2)).reshape(8, 6)
    assert np.array_equal(got['ais'], dbm.ais(12, 7, 1, 777))


def test_shard():
    from boltzmann_machines_amd.parallel import shard
    assert [shard(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert [shard(20000, r, 8) for r in range(8)][-1] == (17500, 20000)
    assert shard(2, 1, 4) == (1, 2) and shard(2, 3, 4) == (2, 2)
