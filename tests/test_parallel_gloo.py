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


class DelayedTwinAsEngine(TwinAsEngine):
    """OracleRBM behind the two-slot interface of DelayedDataParallelRBM"""

    def __init__(self, twin):
        TwinAsEngine.__init__(self, twin)
        self.slots, self.slot = [None, None], 0

    def set_grad_slot(self, slot):
        self.slot = slot

    def grad_step(self, X, B, k):
        self.slots[self.slot] = self.twin.raw_grads(X, k)

    def apply_step(self, B_global, lr, mom):
        self.twin.apply(self.slots[self.slot], float(B_global), lr, mom)


def _delayed_reference(V, H, W, Xg, kw, steps, k, lr, mom):
    """single process, global batch: the same delayed schedule (update t = gradient t-1), then the flush"""
    from oracle import oracle as orc
    ref = orc.OracleRBM(V, H, **kw)
    ref.p['W'][...] = W
    ref.set_seed(99)
    pending = None
    for step in range(steps):
        raw = ref.raw_grads(Xg[step], k)
        if pending is not None:
            ref.apply(pending, float(len(Xg[step])), lr, mom)
        pending = raw
    ref.apply(pending, float(len(Xg[0])), lr, mom)
    return ref


def _delayed_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from boltzmann_machines_amd import parallel
    from oracle import oracle as orc
    dist.init_process_group('gloo', rank=rank, world_size=world)
    V, H, Bl, k, steps = 24, 16, 6, 1, 4
    kw = dict(sample_v_states=True, l2=1e-3)
    W = (orc.normal(1, 2, 0, V * H) * np.float32(0.1)).reshape(V, H)
    twin = orc.OracleRBM(V, H, **kw)
    twin.p['W'][...] = W
    twin.set_seed(99)
    eng = DelayedTwinAsEngine(twin)
    works = [None, None]

    def start_reduce(slot):                      # asynchronous all-reduce of the slot's buffer
        works[slot] = dist.all_reduce(torch.from_numpy(eng.slots[slot]), async_op=True)

    def finish_reduce(slot):
        works[slot].wait()
    dp = parallel.DelayedDataParallelRBM(eng, rank, world, Bl, start_reduce=start_reduce, finish_reduce=finish_reduce)
    for step in range(steps):
        Xg = (orc.uniform(1, 30 + step, 0, world * Bl * V) < 0.3).astype(np.float32).reshape(world * Bl, V)
        dp.train_step(Xg[rank * Bl:(rank + 1) * Bl], 0.05, 0.5, k)
    dp.flush()
    np.savez(out + ('.r%d' % rank), W=twin.p['W'], vb=twin.p['vb'], hb=twin.p['hb'])
    dist.destroy_process_group()


def test_delayed_gradient_dp_world2(tmp_path):
    # the NON-parity mode of DESIGN 6: two ranks running the delayed schedule == one process running the same
    # delayed schedule on the global batch (to the blocking of the sums), and the replicas stay identical
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    out = str(tmp_path / 'delayed')
    mp.spawn(_delayed_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    g0, g1 = np.load(out + '.r0.npz'), np.load(out + '.r1.npz')
    for n in ('W', 'vb', 'hb'):
        assert np.array_equal(g0[n], g1[n])
    V, H, Bl, k, steps, world = 24, 16, 6, 1, 4, 2
    kw = dict(sample_v_states=True, l2=1e-3)
    W = (orc.normal(1, 2, 0, V * H) * np.float32(0.1)).reshape(V, H)
    Xg = [(orc.uniform(1, 30 + s, 0, world * Bl * V) < 0.3).astype(np.float32).reshape(world * Bl, V) for s in range(steps)]
    ref = _delayed_reference(V, H, W, Xg, kw, steps, k, 0.05, 0.5)
    for n in ('W', 'vb', 'hb'):
        np.testing.assert_allclose(g0[n], ref.p[n], rtol=2e-5, atol=2e-7)
    # and it is NOT the synchronous trajectory
    sync = orc.OracleRBM(V, H, **kw)
    sync.p['W'][...] = W
    sync.set_seed(99)
    for s in range(steps):
        sync.train_step(Xg[s], 0.05, 0.5, k)
    assert not np.allclose(g0['W'], sync.p['W'], rtol=1e-6, atol=1e-9)


def test_shard():
    from boltzmann_machines_amd.parallel import shard
    assert [shard(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert [shard(20000, r, 8) for r in range(8)][-1] == (17500, 20000)
    assert shard(2, 1, 4) == (1, 2) and shard(2, 3, 4) == (2, 2)


# ------------------------------------------------------------------ data-parallel DBM incl. the mean-field max
class NumpyDbmAsEngine(object):
    """adapts tests/np_reference.NumpyDBM (float64 restatement of dbm.py) to the engine interface that
    parallel.DataParallelDBM drives: set_row_offset / set_mf_allreduce / grad_step / apply_step"""

    def __init__(self, npm):
        self.m, self.N, self.M = npm, npm.N, npm.M
        self.S = None
        self.keys = None

    def set_row_offset(self, row0, particle0):
        self.m.prow0 = particle0

    def set_mf_allreduce(self, fn):
        self.m.allreduce_max = fn

    def grad_step(self, X, k):
        n_mf, _, self.S = self.m.raw_sums(X, k)
        self.m.call += 1
        return n_mf

    def flat(self):
        self.keys = sorted(self.S)
        return np.concatenate([self.S[k_].ravel() for k_ in self.keys])

    def unflat(self, buf):
        o = 0
        for k_ in self.keys:
            n = self.S[k_].size
            self.S[k_] = buf[o:o + n].reshape(self.S[k_].shape)
            o += n

    def apply_step(self, N_global, M_global, lr, mom):
        self.m.apply(self.S, float(N_global), float(M_global), lr, mom)


def _dbm_model(N, M, rows=None, prow=None):
    """NumpyDBM 14-10-8 with pinned weights; particles / mu for global rows [rows], particles [prow]"""
    from tests import np_reference as ref
    from oracle import oracle as orc
    V, nh = 14, [10, 8]
    n = [V] + nh
    Ng, Mg = 8, 6                                     # global batch / particle count (2 ranks x 4 / 3)
    P = {}
    for i in range(2):
        sfx = '' if i == 0 else '_1'
        P['W' + sfx] = (orc.normal(9, 1 + i, 0, n[i] * n[i + 1]) * np.float32(0.3)).reshape(n[i], n[i + 1]).astype(np.float64)
        P['hb' + sfx] = (orc.uniform(9, 5 + i, 0, n[i + 1]).astype(np.float64) - 0.5) * 0.4
        Hp = (orc.uniform(9, 10 + i, 0, Mg * n[i + 1]) < 0.5).astype(np.float64).reshape(Mg, n[i + 1])
        P['h' + sfx] = Hp[prow].copy()
        for b in ('dW',):
            P[b + sfx] = np.zeros((n[i], n[i + 1]))
        for b in ('dhb', 'q_means', 'mu_means'):
            P[b + sfx] = np.zeros(n[i + 1])
        P['mu' + sfx] = np.zeros((len(range(*rows.indices(Ng))), n[i + 1]))
    P['vb'] = (orc.uniform(9, 20, 0, V).astype(np.float64) - 0.5) * 0.4
    P['dvb'] = np.zeros(V)
    P['v'] = (orc.uniform(9, 21, 0, Mg * V) < 0.3).astype(np.float64).reshape(Mg, V)[prow].copy()
    m = ref.NumpyDBM(P, 2, N, M, max_mf=6, mf_tol=1e-3, l2=1e-3, max_norm=1.2, sp_target=[0.2, 0.1], sp_cost=[1e-2, 5e-3])
    m.seed = 77
    X = (orc.uniform(9, 30, 0, 3 * Ng * V) < 0.25).astype(np.float64).reshape(3, Ng, V)
    return m, X


def _dbm_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from boltzmann_machines_amd import parallel
    dist.init_process_group('gloo', rank=rank, world_size=world)
    N, M = 4, 3
    m, X = _dbm_model(N, M, rows=slice(rank * N, (rank + 1) * N), prow=slice(rank * M, (rank + 1) * M))
    eng = NumpyDbmAsEngine(m)
    calls = []

    def allreduce_():
        t = torch.from_numpy(eng.flat())
        dist.all_reduce(t)
        eng.unflat(t.numpy())

    def allreduce_max(x):
        calls.append(x)
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    dp = parallel.DataParallelDBM(eng, rank, world, allreduce_, allreduce_max=allreduce_max)
    nmf = [dp.train_step(X[s][rank * N:(rank + 1) * N], 0.05, 0.5, 2) for s in range(3)]
    np.savez(out + '.r%d' % rank, nmf=nmf, n_max_calls=len(calls), **{k_: v for k_, v in m.P.items()})
    dist.destroy_process_group()


def test_dp_dbm_world2_with_mean_field_max(tmp_path):
    """DataParallelDBM over gloo, world 2: rank-sharded rows and particles, the mean-field loop condition
    all-reduced (max) per sweep, ONE all-reduce(sum) of the raw sums, update with the global N and M ==
    the same model trained by one process on the concatenated minibatch / particle set."""
    import torch.multiprocessing as mp
    out = str(tmp_path / 'dpdbm')
    mp.spawn(_dbm_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = np.load(out + '.r0.npz'), np.load(out + '.r1.npz')
    ref_m, X = _dbm_model(8, 6, rows=slice(0, 8), prow=slice(0, 6))
    nmf = [ref_m.train_step(X[s], 0.05, 0.5, 2)[0] for s in range(3)]
    assert list(r0['nmf']) == nmf and list(r1['nmf']) == nmf          # the GLOBAL residual decides the trip count
    assert int(r0['n_max_calls']) == int(r1['n_max_calls']) >= sum(nmf)
    for k_ in ('W', 'W_1', 'hb', 'hb_1', 'vb', 'dW', 'q_means', 'mu_means_1'):
        assert np.array_equal(r0[k_], r1[k_]), k_                     # replicas identical
        np.testing.assert_allclose(r0[k_], ref_m.P[k_], rtol=1e-10, atol=1e-13, err_msg=k_)
    # sharded state = slices of the single-process state (sample bitmaps are functions of the GLOBAL row index)
    for k_ in ('v', 'h', 'h_1'):
        assert np.array_equal(np.concatenate([r0[k_], r1[k_]]), ref_m.P[k_]), k_
    for k_ in ('mu', 'mu_1'):
        np.testing.assert_allclose(np.concatenate([r0[k_], r1[k_]]), ref_m.P[k_], rtol=1e-10, atol=1e-13)


class _RecordingDbm(object):
    """the calls DataParallelDBM makes, in order (host logic of the fused / two-step exchange; no device)"""
    N, M = 12, 8

    def __init__(self):
        self.calls = []

    def set_row_offset(self, row0, prow0):
        self.calls.append(('row_offset', row0, prow0))

    def set_xchg(self, x):
        self.calls.append(('set_xchg', x is not None))

    def grad_step(self, X, k, **kw):
        self.calls.append(('grad_step', k, kw.get('row', 0)))
        return 7

    def apply_step(self, N_global, M_global, lr, momentum):
        self.calls.append(('apply_step', N_global, M_global, lr, momentum))


class _RecordingExchange(object):
    def __init__(self, ok):
        self.ok, self.calls = ok, []

    def fused_ok(self):
        return self.ok

    def exchange_apply(self, B_global, lr, momentum, M_global=None):
        self.calls.append(('exchange_apply', B_global, M_global, lr, momentum))


@pytest.mark.parametrize('ok', [True, False])
def test_dp_dbm_host_logic_of_the_fused_exchange(ok):
    """fused: grad_step -> exchange_apply(N * world, lr, momentum, M_global = M * world), no all-reduce, no apply_step;
    an exchange that cannot serve the engine (fused_ok() false: a hidden width that is no multiple of 4) falls back to
    all-reduce + apply_step with the same global sizes; the executed sweep count is returned either way"""
    from boltzmann_machines_amd import parallel
    eng, x, reduced = _RecordingDbm(), _RecordingExchange(ok), []
    dp = parallel.DataParallelDBM(eng, 2, 3, lambda: reduced.append(1), xchg=x, fused=x)
    assert (dp.fused is x) == ok
    assert eng.calls[:2] == [('row_offset', 2 * eng.N, 2 * eng.M), ('set_xchg', True)]
    assert dp.train_step('X', 0.05, 0.5, 4, row=24) == 7
    assert eng.calls[2] == ('grad_step', 4, 24)
    if ok:
        assert x.calls == [('exchange_apply', 3 * eng.N, 3 * eng.M, 0.05, 0.5)] and not reduced and len(eng.calls) == 3
    else:
        assert not x.calls and reduced == [1] and eng.calls[3] == ('apply_step', 3 * eng.N, 3 * eng.M, 0.05, 0.5)
