"""-m gpu: data-parallel training with world_size 2 and 3 ON THE GPU — one process per rank, each with its own
bm_rbm / bm_dbm handle (all on the one GPU of the test box: hipIpc maps another process's allocation whether or
not it lives on another device), the exchange step through the library's direct peer-memory all-reduce
(`bm_xchg_*`: reduce-scatter + all-gather in one kernel per rank, sums in rank order).  RCCL refuses two ranks on
one device; its path is covered at world 1 below and by the driver's multi-GPU bench.  Checked: replicas stay
identical, the update equals the oracle's shard algebra BIT FOR BIT, and the sample bitmaps (functions of the GLOBAL
row) equal a single-process full-batch run."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V, H, BL, K = 100, 52, 24, 2
KW = dict(sample_v_states=True, l2=1e-3, sparsity_cost=1e-2)


def _inputs(world=2):
    from oracle import oracle as orc
    W = (orc.normal(1, 2, 0, V * H) * np.float32(0.1)).reshape(V, H)
    Xg = (orc.uniform(1, 3, 0, world * BL * V) < 0.3).astype(np.float32).reshape(world * BL, V)
    return W, Xg


def _worker(rank, world, port, out, fused=False, cost=1e-2, skew_rank=-1, die_rank=-1):
    sys.path.insert(0, ROOT)
    KW = dict(sample_v_states=True, l2=1e-3, sparsity_cost=cost)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from boltzmann_machines_amd import parallel
    from boltzmann_machines_amd.engine import RbmEngine, as_device
    W, Xg = _inputs(world)
    eng = RbmEngine(V, H, max_batch=BL, **KW)
    eng.set('W', W)
    eng.seed(99)
    # the blobs travel over a plain TCP socket (no process group): the library's exchange needs nothing else
    xchg = parallel.DirectExchange(eng, rank, world, gather=lambda b: parallel.socket_allgather(b, rank, world))
    # fused: the all-reduce AND the update in one kernel per rank (bm_rbm_exchange_apply_direct): reduce-scatter, update of
    # the owned slice of W / dW, all-gather of the new W; every replica updates its own biases from the reduced tail
    dp = parallel.DataParallelRBM(eng, rank, world, BL, parallel.direct_allreduce_on_engine_stream(eng, xchg),
                                  fused=xchg if fused else None)
    assert (dp.fused is not None) == bool(fused)
    Xd = as_device(Xg[rank * BL:(rank + 1) * BL])
    if die_rank >= 0:
        # a rank that disappears mid-run: the survivors' waits must expire (bounded), the status word turn sticky and
        # sync() raise - within the time-out, never a hang and never a silently wrong sum
        import time
        xchg.set_timeout(1.0)
        dp.train_step(Xd, 0.05, 0.5, K)
        eng.sync()
        if rank == die_rank:
            os._exit(0)                          # no clean-up, no farewell: what a crashed process looks like
        t0 = time.time()
        try:
            for step in range(2):
                dp.train_step(Xd, 0.05, 0.5, K)
            eng.sync()
        except RuntimeError as e:
            with open(out + '.r%d.err' % rank, 'w') as f:
                f.write('%.2f %s' % (time.time() - t0, e))
            os._exit(0)
        os._exit(7)                              # two updates "succeeded" without one of the ranks
    for step in range(3):
        if rank == skew_rank:                    # skewed arrivals, what a real fabric adds: 50 ms late every step
            import time
            eng.sync()
            time.sleep(0.05)
        dp.train_step(Xd, 0.05, 0.5, K)
    if fused:
        # between updates a rank holds ITS slice of the momentum buffer: every reader of dW refuses until the replicas
        # are completed (round-4 advisor: a checkpoint or the unfused step would silently use inconsistent momentum)
        for reader in (lambda: eng.get('dW'), lambda: eng.train_step(Xd, BL, 0.05, 0.5, K),
                       lambda: eng.apply_step(world * BL, 0.05, 0.5)):
            try:
                reader()
            except RuntimeError as e:
                assert 'gather_dw' in str(e), e
            else:
                raise AssertionError('dW was read while it is sharded over the ranks')
        eng.get('W')                        # the other variables are whole on every rank
        xchg.gather_dw()                    # ... the replicas are completed on demand
        xchg.gather_dw()                    # (a no-op the second time)
    eng.sync()
    assert xchg.status() == 0
    np.savez(out + '.r%d' % rank, **{n: eng.get(n) for n in ('W', 'vb', 'hb', 'dW', 'q_means')})
    xchg.close()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('world,fused,cost', [(2, False, 1e-2), (3, False, 1e-2), (2, True, 1e-2), (3, True, 1e-2), (2, True, 0.0)])
def test_dp_direct_exchange_on_gpu(gpu_lib, tmp_path, world, fused, cost):
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    KW = dict(sample_v_states=True, l2=1e-3, sparsity_cost=cost)
    out = str(tmp_path / 'dp')
    mp.spawn(_worker, args=(world, _free_port(), out, fused, cost), nprocs=world, join=True)
    rs = [np.load(out + '.r%d.npz' % r) for r in range(world)]
    for r in rs[1:]:
        for n in rs[0].files:                                  # replicas identical
            assert np.array_equal(rs[0][n].view(np.uint32), r[n].view(np.uint32)), n
    # the oracle's shard algebra: raw sums per shard (global row offsets), added in RANK ORDER, applied with N = world*BL
    W, Xg = _inputs(world)
    twins = []
    for r in range(world):
        t = orc.OracleRBM(V, H, **KW)
        t.p['W'][...] = W
        t.set_seed(99)
        t.row0 = r * BL
        twins.append(t)
    for step in range(3):
        raws = [t.raw_grads(Xg[r * BL:(r + 1) * BL], K) for r, t in enumerate(twins)]
        total = raws[0]
        for r in range(1, world):
            total = total + raws[r]
        for t in twins:
            t.apply(total, float(world * BL), 0.05, 0.5)
    for n in ('W', 'vb', 'hb', 'dW', 'q_means'):
        assert np.array_equal(rs[0][n].view(np.uint32), twins[0].p[n].view(np.uint32)), n
    # and the same model trained on the full batch by one engine agrees to fp32 round-off
    # (the per-rank sums are blocked differently), i.e. the rank count only changes the rounding
    ref = orc.OracleRBM(V, H, **KW)
    ref.p['W'][...] = W
    ref.set_seed(99)
    for step in range(3):
        ref.train_step(Xg, 0.05, 0.5, K)
    np.testing.assert_allclose(rs[0]['W'], ref.p['W'], rtol=2e-5, atol=2e-7)


@pytest.mark.parametrize('fused', [False, True])
def test_dp_direct_exchange_with_a_late_rank(gpu_lib, tmp_path, fused):
    """one rank arrives 50 ms late at every exchange (flags, generations and the staging buffers must not care):
    bit-identical to the run without the delay"""
    import torch.multiprocessing as mp
    world = 3
    outs = []
    for tag, skew in (('ontime', -1), ('late', 1)):
        out = str(tmp_path / tag)
        mp.spawn(_worker, args=(world, _free_port(), out, fused, 1e-2, skew), nprocs=world, join=True)
        outs.append([np.load(out + '.r%d.npz' % r) for r in range(world)])
    for r in range(world):
        for n in outs[0][r].files:
            assert np.array_equal(outs[0][r][n].view(np.uint32), outs[1][r][n].view(np.uint32)), (r, n)


@pytest.mark.parametrize('fused', [False, True])
def test_dp_direct_exchange_when_a_rank_dies(gpu_lib, tmp_path, fused):
    """a rank exits without a word after the first update: every survivor's next exchange must fail with an error
    inside the time-out (1 s per wait here) - no hang, no result built on a partial sum"""
    import torch.multiprocessing as mp
    world = 3
    out = str(tmp_path / 'dead')
    mp.spawn(_worker, args=(world, _free_port(), out, fused, 1e-2, -1, 2), nprocs=world, join=True)
    for r in (0, 1):
        with open(out + '.r%d.err' % r) as f:
            secs, msg = f.read().split(' ', 1)
        assert float(secs) < 20.0, secs
        assert 'expired' in msg or 'exchange' in msg, msg


DV, DNH, DN, DM = 36, [24, 16], 12, 8
DKW = dict(max_mf_updates=6, mf_tol=1e-4, l2=1e-3, max_norm=1.5, sparsity_target=[0.2, 0.1], sparsity_cost=[1e-2, 5e-3])


BIG = dict(dims=(784, [512, 1024]), N=512, M=512, scale=0.05,
           kw=dict(max_mf_updates=50, mf_tol=1e-7, l2=1e-7, max_norm=6., sparsity_target=[0.2, 0.1], sparsity_cost=[1e-4, 5e-5]))


def _dbm_setup(rows, prow, N, M, dims=None, scale=0.2, kw=None, Nl=None, Ml=None):
    """DbmEngine with pinned parameters; the particles of global indices [prow].  N, M: the engine's rows / particles;
    Nl, Ml: rows / particles of ONE rank of the 2-rank run (the global arrays have twice as many)"""
    from boltzmann_machines_amd.engine import DbmEngine
    from oracle import oracle as orc
    dv, dnh = dims or (DV, DNH)
    Nl, Ml = Nl or DN, Ml or DM
    n = [dv] + list(dnh)
    eng = DbmEngine(dv, list(dnh), n_particles=M, batch_size=N, **(kw or DKW))
    Mg = 2 * Ml
    for i in range(2):
        sfx = '' if i == 0 else '_1'
        eng.set('W' + sfx, (orc.normal(5, 1 + i, 0, n[i] * n[i + 1]) * np.float32(scale)).reshape(n[i], n[i + 1]))
        eng.set('hb' + sfx, (orc.uniform(5, 5 + i, 0, n[i + 1]) - np.float32(0.5)) * np.float32(0.4))
        eng.set('h' + sfx, (orc.uniform(5, 10 + i, 0, Mg * n[i + 1]) < 0.5).astype(np.float32).reshape(Mg, n[i + 1])[prow])
    eng.set('v', (orc.uniform(5, 21, 0, Mg * dv) < 0.3).astype(np.float32).reshape(Mg, dv)[prow])
    X = (orc.uniform(5, 30, 0, 3 * 2 * Nl * dv) < 0.25).astype(np.float32).reshape(3, 2 * Nl, dv)
    eng.seed(77)
    return eng, X[:, rows]


def _dbm_worker(rank, world, port, out, big=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from boltzmann_machines_amd import parallel
    from boltzmann_machines_amd.engine import as_device
    if big:
        n_, m_ = BIG['N'], BIG['M']
        eng, X = _dbm_setup(slice(rank * n_, (rank + 1) * n_), slice(rank * m_, (rank + 1) * m_), n_, m_, BIG['dims'],
                            BIG['scale'], BIG['kw'], n_, m_)
    else:
        eng, X = _dbm_setup(slice(rank * DN, (rank + 1) * DN), slice(rank * DM, (rank + 1) * DM), DN, DM)
    # (two ranks on one device: the exchange must not spin on every CU - bm_xchg_set_max_workgroups)
    xchg = parallel.DirectExchange(eng, rank, world, gather=lambda b: parallel.socket_allgather(b, rank, world), max_workgroups=48)
    dp = parallel.DataParallelDBM(eng, rank, world, parallel.direct_allreduce_on_engine_stream(eng, xchg), xchg=xchg,
                                  fused=xchg if big else None)
    nmf, first = [], None
    for s in range(3):
        nmf.append(dp.train_step(as_device(X[s]), 0.05, 0.5, 2))
        if s == 0:
            first = {k_: eng.get(k_) for k_ in ('v', 'h', 'h_1', 'W', 'W_1', 'hb', 'hb_1', 'vb')}
    if big:
        xchg.gather_dw()
    eng.sync()
    assert xchg.status() == 0
    names = ('W', 'W_1', 'hb', 'hb_1', 'vb', 'dW', 'q_means', 'mu_means_1')
    np.savez(out + '.r%d' % rank, nmf=nmf, **{k_: eng.get(k_) for k_ in names}, **{'first_' + k_: a for k_, a in first.items()})
    eng.set_xchg(None)
    xchg.close()


@pytest.mark.parametrize('big', [False, True])
def test_dp_dbm_direct_exchange_on_gpu(gpu_lib, tmp_path, big):
    """DataParallelDBM, world 2, both exchange steps over the direct path: the mean-field residual max per sweep
    (bm_xchg_allreduce_max1, device side, in stream order) and the all-reduce(sum) of the fused gradient buffer.
    Replicas identical; the executed sweep counts are the global ones (= a single engine on the concatenated
    minibatch); the particles after the first update are the slices of that engine's particles bit for bit;
    parameters agree with it to fp32 round-off (the shard sums are blocked differently).
    big: BASELINE configs[3] per rank - 784-512-1024, 512 rows + 512 particles per rank, up to 50 sweeps at tol 1e-7, the
    fused column-sliced exchange - against ONE engine with 1024 rows + 1024 particles."""
    import torch.multiprocessing as mp
    from boltzmann_machines_amd.engine import as_device
    out = str(tmp_path / 'dpdbm')
    mp.spawn(_dbm_worker, args=(2, _free_port(), out, big), nprocs=2, join=True)
    r0, r1 = np.load(out + '.r0.npz'), np.load(out + '.r1.npz')
    names = ('W', 'W_1', 'hb', 'hb_1', 'vb', 'dW', 'q_means', 'mu_means_1')
    for k_ in names:
        assert np.array_equal(r0[k_].view(np.uint32), r1[k_].view(np.uint32)), k_
    if big:
        n_, m_ = BIG['N'], BIG['M']
        ref, X = _dbm_setup(slice(0, 2 * n_), slice(0, 2 * m_), 2 * n_, 2 * m_, BIG['dims'], BIG['scale'], BIG['kw'], n_, m_)
    else:
        ref, X = _dbm_setup(slice(0, 2 * DN), slice(0, 2 * DM), 2 * DN, 2 * DM)
    nmf = []
    for s in range(3):
        nmf.append(ref.train_step(as_device(X[s]), 0.05, 0.5, 2)[0])
        if s == 0:
            for k_ in ('v', 'h', 'h_1'):
                both = np.concatenate([r0['first_' + k_], r1['first_' + k_]])
                assert np.array_equal(both.view(np.uint32), ref.get(k_).view(np.uint32)), k_
            for k_ in ('W', 'W_1', 'hb', 'hb_1', 'vb'):     # one update: same particles and mu, only the blocking of the sums
                np.testing.assert_allclose(r0['first_' + k_], ref.get(k_), rtol=5e-5, atol=1e-6, err_msg='first ' + k_)
    print('executed mean-field sweeps', list(r0['nmf']), nmf)
    if big:
        # at tol 1e-7 the trip count hangs on the last ulps of the residual, and from the second update on the replicas'
        # weights differ from the single engine's by the blocking of the shard sums: the FIRST count must agree exactly
        assert list(r0['nmf']) == list(r1['nmf']) and int(r0['nmf'][0]) == nmf[0], (list(r0['nmf']), list(r1['nmf']), nmf)
        assert nmf[0] > 3
    else:
        assert list(r0['nmf']) == list(r1['nmf']) == nmf, (list(r0['nmf']), list(r1['nmf']), nmf)
    for k_ in names:
        # (big: after three updates a handful of the ~1e7 draws may have flipped on ulp-level weight differences; a flipped
        # bit moves one gradient entry by lr / 1024)
        np.testing.assert_allclose(r0[k_], ref.get(k_), rtol=5e-5, atol=1e-6 if not big else 5e-4, err_msg=k_)
    ref.close()


# ---- chain-sharded AIS over the direct exchange (bm_dbm_ais_sharded_direct; SURVEY 8e, dbm.py:922-939)
AIS_R, AIS_NB, AIS_K, AIS_SEED = 37, 25, 2, 4242          # 37 chains: uneven slices at world 2 (19 + 18) and 3 (13 + 12 + 12)


def _ais_engine(small_buffer=False):
    from boltzmann_machines_amd.engine import DbmEngine
    from oracle import oracle as orc
    dv, dnh = (8, [4, 4]) if small_buffer else (DV, DNH)     # small_buffer: the gradient payload (< 37 floats per window ...
    n = [dv] + list(dnh)
    eng = DbmEngine(dv, list(dnh), n_particles=4, batch_size=4)
    for i in range(2):
        sfx = '' if i == 0 else '_1'
        eng.set('W' + sfx, (orc.normal(5, 1 + i, 0, n[i] * n[i + 1]) * np.float32(0.2)).reshape(n[i], n[i + 1]))
        eng.set('hb' + sfx, (orc.uniform(5, 5 + i, 0, n[i + 1]) - np.float32(0.5)) * np.float32(0.4))
    eng.set('vb', (orc.uniform(5, 9, 0, dv) - np.float32(0.5)) * np.float32(0.4))
    return eng


def _ais_worker(rank, world, port, out, die_rank=-1, n_runs=AIS_R):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import time
    from boltzmann_machines_amd import parallel
    eng = _ais_engine()
    xchg = parallel.DirectExchange(eng, rank, world, gather=lambda b: parallel.socket_allgather(b, rank, world), max_workgroups=48)
    if die_rank >= 0:
        xchg.set_timeout(1.0)
        v0 = eng.ais_sharded_direct(xchg, AIS_NB, n_runs, AIS_K, AIS_SEED)        # one good round first
        if rank == die_rank:
            os._exit(0)
        t0, err = time.time(), ''
        try:
            eng.ais_sharded_direct(xchg, AIS_NB, n_runs, AIS_K, AIS_SEED)
        except Exception as e:       # noqa: BLE001
            err = str(e)
        np.savez(out + '.r%d' % rank, v0=v0, seconds=time.time() - t0, err=np.array(err))
        os._exit(0)                                                               # (the collective close would wait for the dead rank)
    vals = [eng.ais_sharded_direct(xchg, AIS_NB, n_runs, AIS_K, AIS_SEED) for _ in range(2)]   # twice: warm flags / staging
    # and a run with more chains than one window of a tiny registered buffer would hold is covered by the window loop:
    eng.sync()
    assert xchg.status() == 0
    np.savez(out + '.r%d' % rank, v0=vals[0], v1=vals[1])
    xchg.close()


@pytest.mark.parametrize('world', [2, 3])
def test_ais_sharded_direct(gpu_lib, tmp_path, world):
    """every rank returns ALL chain values, and they are the single-engine run's values BIT FOR BIT (a chain's RNG stream is
    addressed by its global index; the exchange adds zeros to the owner's value)"""
    import torch.multiprocessing as mp
    out = str(tmp_path / 'ais')
    mp.spawn(_ais_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    eng = _ais_engine()
    ref = eng.ais(AIS_NB, AIS_R, AIS_K, AIS_SEED)
    eng.close()
    assert np.all(np.isfinite(ref))
    for r in range(world):
        z = np.load(out + '.r%d.npz' % r)
        for k_ in ('v0', 'v1'):
            assert np.array_equal(z[k_].view(np.uint32), ref.view(np.uint32)), (r, k_)


def test_ais_sharded_direct_when_a_rank_dies(gpu_lib, tmp_path):
    """a rank that disappears between two runs: the survivor's wait expires at the exchange's time-out and the call RAISES
    (sticky status, NaN-poisoned window) - bounded, never a hang and never a silently partial result"""
    import torch.multiprocessing as mp
    out = str(tmp_path / 'aisdie')
    mp.spawn(_ais_worker, args=(2, _free_port(), out, 1), nprocs=2, join=True)
    z = np.load(out + '.r0.npz')
    assert np.all(np.isfinite(z['v0']))
    assert str(z['err']) != '' and float(z['seconds']) < 30.0, (str(z['err']), float(z['seconds']))


def test_ais_sharded_direct_in_windows(gpu_lib):
    """world 1, a registered buffer SHORTER than the number of chains (8-4-4 DBM: 2 * (32 + 16) + sums floats): the values
    travel in several windows and equal bm_dbm_ais"""
    from boltzmann_machines_amd import parallel
    eng = _ais_engine(small_buffer=True)
    xchg = parallel.DirectExchange(eng, 0, 1)
    n = eng.device_view('grad').shape[0]
    R = 2 * n + 5
    got = eng.ais_sharded_direct(xchg, 6, R, 1, 99)
    ref = eng.ais(6, R, 1, 99)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    xchg.close(); eng.close()


# ---- the fused DBM exchange (bm_dbm_exchange_apply_direct): column-sliced ownership.  Widths 96 and 64: at world 2 the
# slices are 64 + 32 and 32 + 32 columns, at world 3 they are 32 + 32 + 32 and 32 + 32 + NONE (a rank without columns of W_1)
FV, FNH, FN, FM = 40, [96, 64], 12, 8
FKW = dict(max_mf_updates=6, mf_tol=1e-4, l2=1e-3, max_norm=1.5, sparsity_target=[0.2, 0.1], sparsity_cost=[1e-2, 5e-3])
FNAMES = ('W', 'W_1', 'hb', 'hb_1', 'vb', 'dvb', 'dW', 'dW_1', 'dhb', 'dhb_1', 'q_means', 'q_means_1', 'mu_means', 'mu_means_1',
          'W_norm', 'W_norm_1', 'v', 'h', 'h_1', 'mu', 'mu_1')


def _fdbm_setup(world, rank):
    from boltzmann_machines_amd.engine import DbmEngine
    from oracle import oracle as orc
    n = [FV] + FNH
    eng = DbmEngine(FV, FNH, n_particles=FM, batch_size=FN, **FKW)
    Mg = world * FM
    prow = slice(rank * FM, (rank + 1) * FM)
    for i in range(2):
        sfx = '' if i == 0 else '_1'
        eng.set('W' + sfx, (orc.normal(6, 1 + i, 0, n[i] * n[i + 1]) * np.float32(0.2)).reshape(n[i], n[i + 1]))
        eng.set('hb' + sfx, (orc.uniform(6, 5 + i, 0, n[i + 1]) - np.float32(0.5)) * np.float32(0.4))
        eng.set('h' + sfx, (orc.uniform(6, 10 + i, 0, Mg * n[i + 1]) < 0.5).astype(np.float32).reshape(Mg, n[i + 1])[prow])
    eng.set('v', (orc.uniform(6, 21, 0, Mg * FV) < 0.3).astype(np.float32).reshape(Mg, FV)[prow])
    X = (orc.uniform(6, 30, 0, 4 * world * FN * FV) < 0.25).astype(np.float32).reshape(4, world * FN, FV)
    eng.seed(78)
    return eng, X[:, rank * FN:(rank + 1) * FN]


def _fdbm_worker(rank, world, port, out, fused, skew_rank=-1, die_rank=-1):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from boltzmann_machines_amd import parallel
    from boltzmann_machines_amd.engine import as_device
    eng, X = _fdbm_setup(world, rank)
    xchg = parallel.DirectExchange(eng, rank, world, gather=lambda b: parallel.socket_allgather(b, rank, world))
    xchg.set_timeout(5.0)
    dp = parallel.DataParallelDBM(eng, rank, world, parallel.direct_allreduce_on_engine_stream(eng, xchg), xchg=xchg,
                                  fused=xchg if fused else None)
    assert (dp.fused is not None) == bool(fused)
    nmf = []
    if die_rank >= 0:
        import time
        xchg.set_timeout(1.0)
        dp.train_step(as_device(X[0]), 0.05, 0.5, 2)
        eng.sync()
        if rank == die_rank:
            os._exit(0)
        t0 = time.time()
        try:
            for s in range(1, 3):
                dp.train_step(as_device(X[s]), 0.05, 0.5, 2)
            eng.sync()
        except RuntimeError as e:
            with open(out + '.r%d.err' % rank, 'w') as f:
                f.write('%.2f %s' % (time.time() - t0, e))
            os._exit(0)
        os._exit(7)
    for s in range(4):
        if rank == skew_rank:
            import time
            eng.sync()
            time.sleep(0.05)
        nmf.append(dp.train_step(as_device(X[s]), 0.05, 0.5, 2))
        if fused and s == 1 and world > 1 and skew_rank < 0:
            # the momentum buffers are column-sharded between updates: every reader refuses until they are gathered,
            # and a gather in the middle of a run changes nothing
            for reader in (lambda: eng.get('dW'), lambda: eng.get('dW_1'), lambda: eng.train_step(as_device(X[s]), 0.05, 0.5, 2),
                           lambda: eng.apply_step(world * FN, world * FM, 0.05, 0.5)):
                try:
                    reader()
                except RuntimeError as e:
                    assert 'gather_dw' in str(e), e
                else:
                    raise AssertionError('dW was read while it is sharded over the ranks')
            eng.get('W_1')
            xchg.gather_dw()
            eng.get('dW_1')
    if fused:
        xchg.gather_dw()
        xchg.gather_dw()
    eng.sync()
    assert xchg.status() == 0
    np.savez(out + '.r%d' % rank, nmf=nmf, **{k_: eng.get(k_) for k_ in FNAMES})
    eng.set_xchg(None)
    xchg.close()


@pytest.mark.parametrize('world', [2, 3])
def test_dp_dbm_fused_exchange_on_gpu(gpu_lib, tmp_path, world):
    """bm_dbm_exchange_apply_direct against bm_dbm_allreduce_grads_direct + bm_dbm_apply_step: four updates (so that the
    gathered W_i^T, biases, penalties and particles of one update feed the next), every variable BIT FOR BIT - weights,
    momentum buffers after gather_dw, column norms, running means, particles, mean-field parameters, sweep counts -
    replicas identical, including a rank that owns no column of a layer (world 3)."""
    import torch.multiprocessing as mp
    outs = []
    for tag, fused in (('two_step', False), ('fused', True)):
        out = str(tmp_path / tag)
        mp.spawn(_fdbm_worker, args=(world, _free_port(), out, fused), nprocs=world, join=True)
        outs.append([np.load(out + '.r%d.npz' % r) for r in range(world)])
    shared = [n for n in FNAMES if n not in ('v', 'h', 'h_1', 'mu', 'mu_1')]
    for r in range(world):
        assert list(outs[0][r]['nmf']) == list(outs[1][r]['nmf']) == list(outs[1][0]['nmf'])
        for n in FNAMES:
            assert np.array_equal(outs[0][r][n].view(np.uint32), outs[1][r][n].view(np.uint32)), (r, n)
        for n in shared:
            assert np.array_equal(outs[1][0][n].view(np.uint32), outs[1][r][n].view(np.uint32)), ('replicas', r, n)
    assert np.all(np.isfinite(outs[1][0]['W_1'])) and float(np.abs(outs[1][0]['dW_1']).max()) > 0


def test_dp_dbm_fused_exchange_with_a_late_and_with_a_dying_rank(gpu_lib, tmp_path):
    """the column-sliced exchange under skewed arrivals (one rank 50 ms late at every update: same bits as on time) and
    with a rank that exits after the first update (every survivor fails inside the time-out; nobody hangs)"""
    import torch.multiprocessing as mp
    world = 3
    outs = []
    for tag, skew in (('ontime', -1), ('late', 2)):
        out = str(tmp_path / tag)
        mp.spawn(_fdbm_worker, args=(world, _free_port(), out, True, skew), nprocs=world, join=True)
        outs.append([np.load(out + '.r%d.npz' % r) for r in range(world)])
    for r in range(world):
        for n in FNAMES:
            assert np.array_equal(outs[0][r][n].view(np.uint32), outs[1][r][n].view(np.uint32)), (r, n)
    out = str(tmp_path / 'dead')
    mp.spawn(_fdbm_worker, args=(world, _free_port(), out, True, -1, 1), nprocs=world, join=True)
    for r in (0, 2):
        with open(out + '.r%d.err' % r) as f:
            secs, msg = f.read().split(' ', 1)
        # (two updates of up to 7 loop-control exchanges + 2 launches each, 1 s per expired wait: ~18 s; the bound only says
        # "no hang")
        assert float(secs) < 60.0, secs
        assert 'expired' in msg or 'exchange' in msg, msg


NATIVE_SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r)
import ctypes as C
import numpy as np
assert 'torch' not in sys.modules
from boltzmann_machines_amd import _ffi, parallel
from boltzmann_machines_amd._ffi import DeviceArray
from boltzmann_machines_amd.engine import RbmEngine, as_device
from tests.test_parallel_gpu import _inputs, V, H, BL, K, KW
W, Xg = _inputs()
comm = parallel.NativeComm(0, 1, parallel.NativeComm.unique_id())
e1 = RbmEngine(V, H, max_batch=BL, **KW)
e2 = RbmEngine(V, H, max_batch=BL, **KW)
for e in (e1, e2):
    e.set('W', W); e.seed(7)
dp = parallel.DataParallelRBM(e2, 0, 1, BL, parallel.native_allreduce_on_engine_stream(e2, comm))
Xd = as_device(Xg[:BL])
for step in range(3):
    e1.train_step(Xd, BL, 0.05, 0.5, K)
    dp.train_step(Xd, 0.05, 0.5, K)
for n in ('W', 'vb', 'hb', 'dW', 'q_means'):
    assert np.array_equal(e1.get(n).view(np.uint32), e2.get(n).view(np.uint32)), n
a = DeviceArray.from_numpy(np.arange(10, dtype=np.float32))
b = DeviceArray((10,))
_ffi.check(_ffi.load().bm_comm_allgather(comm._c, a.ptr, b.ptr, 10, C.c_void_p(e2.stream())))
e2.sync()
assert np.array_equal(b.numpy(), np.arange(10, dtype=np.float32))
comm.close(); e1.close(); e2.close()
assert 'torch' not in sys.modules
print('NATIVE_COMM_OK')
"""


def test_native_rccl_comm_world1(gpu_lib):
    """the library's own RCCL communicator (bm_comm_*), world 1 on the test box, in a process that never
    imports torch: init, the all-reduce of the fused grad buffer on the engine stream between grad_step and
    apply_step (== the fused train_step bit for bit), all-gather, destroy."""
    import subprocess
    r = subprocess.run([sys.executable, '-c', NATIVE_SCRIPT % dict(root=ROOT)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and 'NATIVE_COMM_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


NATIVE_DBM_SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
assert 'torch' not in sys.modules
from boltzmann_machines_amd import parallel
from boltzmann_machines_amd.engine import as_device
from tests import test_dbm_parity_gpu as D
V, nh, N, M = 20, [12, 16], 10, 10
kw = dict(max_mf_updates=5, mf_tol=1e-5, l2=1e-3, max_norm=1.5, sparsity_target=[0.2, 0.1], sparsity_cost=[1e-2, 5e-3])
comm = parallel.NativeComm(0, 1, parallel.NativeComm.unique_id())
e1, _ = D.make_pair(V, nh, N, M, **kw)
e2, _ = D.make_pair(V, nh, N, M, **kw)
e1.seed(42); e2.seed(42)
dp = parallel.DataParallelDBM(e2, 0, 1, parallel.native_allreduce_on_engine_stream(e2, comm), comm=comm)
for s in range(3):
    Xd = as_device(D.data(N, V, s))
    n1, _ = e1.train_step(Xd, 0.05, 0.5, 2)
    n2 = dp.train_step(Xd, 0.05, 0.5, 2)      # MF residual all-reduced (max) on the device, grads all-reduced (sum)
    assert n1 == n2, (n1, n2)
for nm in ('W', 'W_1', 'dW', 'hb', 'hb_1', 'vb', 'q_means', 'mu_means_1', 'v', 'h_1', 'mu', 'mu_1'):
    assert np.array_equal(e1.get(nm).view(np.uint32), e2.get(nm).view(np.uint32)), nm
# validation fetch and inference also run with the communicator installed
assert e1.metrics(Xd, 2) == e2.metrics(Xd, 2)
# chain-sharded AIS: shard + all-gather inside the library == the plain run
a = e1.ais(20, 37, 1, 2222)
b = e2.ais_sharded(comm, 20, 37, 1, 2222)
assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
e2.set_comm(None)
comm.close(); e1.close(); e2.close()
assert 'torch' not in sys.modules
print('NATIVE_DBM_OK')
"""


def test_native_comm_dbm_and_ais_world1(gpu_lib):
    """DBM data-parallel step and chain-sharded AIS through the library's own RCCL communicator (world 1 on the
    test box, torch-free process): device-side all-reduce(max) of the mean-field residual per sweep, one
    all-reduce(sum) of the fused gradient buffer, one all-gather of the AIS values — bit-identical to the
    fused single-GPU entry points."""
    import subprocess
    r = subprocess.run([sys.executable, '-c', NATIVE_DBM_SCRIPT % dict(root=ROOT)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and 'NATIVE_DBM_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


DELAYED_SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
assert 'torch' not in sys.modules
from boltzmann_machines_amd import parallel
from boltzmann_machines_amd.engine import RbmEngine, as_device
from oracle import oracle as orc
V, H, B, k, steps = 48, 40, 24, 1, 5
kw = dict(l2=1e-3, sample_v_states=True, sample_h_states=True)
W = (orc.normal(1, 2, 0, V * H) * np.float32(0.1)).reshape(V, H)
eng = RbmEngine(V, H, max_batch=B, **kw)
twin = orc.OracleRBM(V, H, **kw)
eng.set('W', W); twin.p['W'][...] = W
eng.seed(5); twin.set_seed(5)
comm = parallel.NativeComm(0, 1, parallel.NativeComm.unique_id())
dp = parallel.DelayedDataParallelRBM(eng, 0, 1, B, comm=comm)
pending = None
for s in range(steps):
    X = (orc.uniform(1, 50 + s, 0, B * V) < 0.3).astype(np.float32).reshape(B, V)
    dp.train_step(as_device(X), 0.05, 0.5, k)
    raw = twin.raw_grads(X, k)                 # the oracle on the same delayed schedule
    if pending is not None:
        twin.apply(pending, float(B), 0.05, 0.5)
    pending = raw
dp.flush(); twin.apply(pending, float(B), 0.05, 0.5)
for nm in ('W', 'dW', 'vb', 'hb', 'dvb', 'dhb', 'q_means'):
    assert np.array_equal(eng.get(nm).view(np.uint32), twin.p[nm].view(np.uint32)), nm
comm.close(); eng.close()
print('DELAYED_OK')
"""


def test_delayed_gradient_dp_world1(gpu_lib):
    """the NON-parity delayed-gradient mode through the library (two gradient slots, the reduction on the
    communication stream, bm_rbm_wait_grads): bit-identical to the oracle driven on the same delayed schedule"""
    import subprocess
    r = subprocess.run([sys.executable, '-c', DELAYED_SCRIPT % dict(root=ROOT)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and 'DELAYED_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
