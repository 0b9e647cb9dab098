"""CPU tests (-m "not gpu"): host logic that mirrors the reference without touching
the device — constructor validation, working paths, utilities, RNG, the W-init known
answer, and that the C-ABI library loads and exports every declared symbol."""
import doctest
import json
import os
import re

import numpy as np
import pytest
from numpy.testing import assert_almost_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_w_init_shape_validation():
    """reference rbm/tests/test_rbm.py:29-36"""
    from boltzmann_machines_amd import BernoulliRBM, GaussianRBM, MultinomialRBM
    for C in (BernoulliRBM, MultinomialRBM, GaussianRBM):
        for bad in ((4, 2), (3, 3), (3, 2)):
            with pytest.raises(ValueError):
                C(n_visible=4, n_hidden=3, W_init=np.zeros(bad))
        C(n_visible=4, n_hidden=3, W_init=np.zeros((4, 3)))
        C(n_visible=3, n_hidden=3, W_init=np.zeros((3, 3)))
        C(n_visible=1, n_hidden=1, W_init=np.zeros((1, 1)))
        with pytest.raises(ValueError):
            C(n_visible=4, n_hidden=3, vb_init=np.zeros(5))
        with pytest.raises(ValueError):
            C(n_visible=4, n_hidden=3, hb_init=np.zeros(5))
        with pytest.raises(AttributeError):
            C(n_visible=4, n_hidden=3, no_such_kwarg=1)      # base/mixin.py:9-10


def test_w_init_known_answer():
    """reference rbm/tests/test_rbm.py:64-67: random_seed=1337, stddev 0.01."""
    from boltzmann_machines_amd import BernoulliRBM
    w32 = BernoulliRBM(n_visible=12, n_hidden=8, random_seed=1337)._initial_variables()['W']
    w64 = BernoulliRBM(n_visible=12, n_hidden=8, random_seed=1337, dtype='float64')._initial_variables()['W']
    assert w32.dtype == np.float32 and w64.dtype == np.float64
    assert_almost_equal(w32[0][0], -0.0094548017)
    assert_almost_equal(w64[0][0], -0.0077341544416)


PATHS = [   # reference base/tests/test_tf_model.py:8-93
    ('model/', ('model/', 'model/model')),
    ('model/my_model', ('model/', 'model/my_model')),
    ('a/b/c/', ('a/b/c/', 'a/b/c/model')),
    ('my_model', ('./', './my_model')),
    ('./', ('./', './model')),
]


@pytest.mark.parametrize('model_path,expected', PATHS)
def test_working_paths(model_path, expected):
    from boltzmann_machines_amd.base import EngineModel
    p = EngineModel.compute_working_paths(model_path)
    assert p['model_dirpath'] == expected[0]
    assert p['model_filepath'] == expected[1]
    assert p['params_filepath'] == expected[0] + 'params.json'
    assert p['random_state_filepath'] == expected[0] + 'random_state.json'
    assert p['train_summary_dirpath'] == expected[0] + 'logs/train'
    assert p['tf_meta_graph_filepath'] == expected[1] + '.meta'


def test_params_roundtrip():
    from boltzmann_machines_amd import BernoulliRBM
    rbm = BernoulliRBM(n_visible=4, n_hidden=3, learning_rate=[0.1, 0.01], random_seed=3)
    p = rbm.get_params()
    assert p['learning_rate'] == [0.1, 0.01] and p['epoch_'] == 0 and 'n_visible' in p
    assert not any(k.startswith('_') for k in p)
    rbm.set_params(max_epoch=7, epoch_=2)
    assert rbm.max_epoch == 7 and rbm.epoch_ == 2
    with pytest.raises(ValueError):
        rbm.set_params(_rng=None)
    json.dumps(rbm._serialize(rbm.get_params(deep=True)))     # params.json payload is serialisable
    s1, s2 = BernoulliRBM(random_seed=9).make_random_seed(), BernoulliRBM(random_seed=9).make_random_seed()
    assert s1 == s2


def test_utils_doctests():
    from boltzmann_machines_amd.utils import rng, utils
    for m in (utils, rng):
        assert doctest.testmod(m).failed == 0


def test_batch_and_epoch_iter():
    from boltzmann_machines_amd.utils import batch_iter, epoch_iter, make_list_from
    X = np.arange(36).reshape(12, 3)
    assert [b.shape[0] for b in batch_iter(X, 5)] == [5, 5, 2]
    assert list(epoch_iter(2, 5)) == [3, 4, 5]
    assert make_list_from(3) == [3] and make_list_from((1, 2)) == [1, 2]


def test_host_philox_matches_oracle():
    from boltzmann_machines_amd.utils import philox
    from oracle import oracle as orc
    assert np.array_equal(philox.philox_blocks(123456789012, 7, 9, 5, 33), orc.philox_words(123456789012, 7, 9, 5, 33))
    assert np.array_equal(philox.uniform(5, 3, 7, 1001, idx0=3), orc.uniform(5, 3, 7, 1001, idx0=3))
    # the host initialiser calls numpy's log / sin / cos, the engine and the oracle the bit-pinned Box-Muller
    # (csrc/bm_rng.h pin_log_unit / pin_sincos_2pi): a few ulp apart, as TF's own libm calls are
    np.testing.assert_allclose(philox.normal(5, 3, 7, 1001), orc.normal(5, 3, 7, 1001), rtol=1e-6, atol=5e-7)


def test_c_abi_exports_every_declared_symbol():
    """libbm355.so loads without a GPU and exports exactly what include/bm355.h declares."""
    from boltzmann_machines_amd import _ffi
    lib = _ffi.load()
    header = open(os.path.join(ROOT, 'include', 'bm355.h')).read()
    declared = set(re.findall(r'\b(bm_[a-z0-9_]+)\s*\(', header))
    declared -= {'bm_rbm_config', 'bm_dbm_config'}
    assert declared, 'no declarations parsed'
    for name in sorted(declared):
        assert hasattr(lib, name), 'libbm355.so does not export %s' % name
    assert declared == set(_ffi.exported_symbols()), declared ^ set(_ffi.exported_symbols())
    assert lib.bm_version().startswith(b'bm355')


def test_no_cpu_fallback_without_gpu():
    """Without a visible GPU the product path must fail loudly, not compute on the CPU."""
    from boltzmann_machines_amd import _ffi
    lib = _ffi.load()
    if lib.bm_device_count() > 0:
        pytest.skip('GPU visible: covered by the -m gpu tests')
    from boltzmann_machines_amd.engine import RbmEngine
    with pytest.raises(_ffi.Bm355Error, match='no HIP device'):
        RbmEngine(8, 4, max_batch=2)
    from boltzmann_machines_amd import BernoulliRBM
    with pytest.raises(_ffi.Bm355Error):
        BernoulliRBM(n_visible=8, n_hidden=4, verbose=False, model_path='/tmp/bm355_nogpu/').fit(np.zeros((4, 8)))


def test_dataset_readers(tmp_path):
    """utils/dataset.py (reference utils/dataset.py:10-130): IDX / CIFAR-pickle readers on synthetic files in the
    reference's directory layout, and the flatten round trips of its doctests."""
    import pickle
    import struct
    from boltzmann_machines_amd.utils import dataset
    rng = np.random.RandomState(0)
    d = tmp_path / 'mnist'
    d.mkdir()
    pix = rng.randint(0, 256, size=(7, 28, 28)).astype(np.uint8)
    lab = rng.randint(0, 10, size=7).astype(np.int8)
    for stem in ('train', 't10k'):
        (d / (stem + '-images-idx3-ubyte')).write_bytes(struct.pack('>IIII', 2051, 7, 28, 28) + pix.tobytes())
        (d / (stem + '-labels-idx1-ubyte')).write_bytes(struct.pack('>II', 2049, 7) + lab.tobytes())
    X, y = dataset.load_mnist('train', str(tmp_path))
    assert X.shape == (7, 784) and X.dtype == np.float64 and np.array_equal(X, pix.reshape(7, -1)) and np.array_equal(y, lab)
    assert dataset.load_mnist('test', str(tmp_path))[0].shape == (7, 784)
    with pytest.raises(ValueError):
        dataset.load_mnist('val', str(tmp_path))
    c = tmp_path / 'cifar-10-batches-py'
    c.mkdir()
    for i, name in enumerate(['data_batch_%d' % k for k in range(1, 6)] + ['test_batch']):
        with open(str(c / name), 'wb') as f:
            pickle.dump({'data': np.full((3, 3072), i, dtype=np.uint8), 'labels': [i, i, i]}, f)
    X, y = dataset.load_cifar10('train', str(tmp_path))
    assert X.shape == (15, 3072) and y.tolist() == sum(([i] * 3 for i in range(5)), [])
    assert dataset.load_cifar10('test', str(tmp_path))[1].tolist() == [5, 5, 5]
    for shape in ((10, 3072), (3072,), (9, 8 * 8 * 3)):
        A = rng.rand(*shape)
        np.testing.assert_allclose(A, dataset.im_flatten(dataset.im_unflatten(A)))
    for shape in ((7, 32, 32, 3), (32, 32, 3), (8, 8, 3)):
        A = rng.rand(*shape)
        np.testing.assert_allclose(A, dataset.im_unflatten(dataset.im_flatten(A)))


def test_generated_isa_has_no_implicit_m0_reader(tmp_path):
    """csrc/bm_gemm.h dma16s writes M0 (the LDS-DMA base) without restoring it - legal only while nothing hipcc
    generates reads M0 implicitly.  Compile the device code to assembly and look."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    from boltzmann_machines_amd import build as b
    flags = [f for f in b.FLAGS if f not in ('-shared', '-fPIC')]
    out = str(tmp_path / 'bm355.s')
    r = subprocess.run([hipcc] + flags + ['--cuda-device-only', '-S', os.path.join(b.CSRC, 'bm355.hip'), '-o', out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'bm_' not in r.stderr, r.stderr[-2000:]               # no diagnostics from the sources themselves
    asm = open(out).read()
    assert 'global_load_lds_dwordx4' in asm                      # the DMA path is really there
    for op in ('s_movrel', 'v_movrel', 's_sendmsg', 'ds_gws', 'v_interp', 'lds_direct', '_addtid'):
        assert op not in asm, 'hipcc emitted %s: it reads M0, which dma16s leaves modified' % op


class _StubModel(object):
    """smallest EngineModel that can save: the variables are host arrays (no device needed)"""

    @staticmethod
    def make(tmp_path):
        from boltzmann_machines_amd.base import EngineModel

        class Stub(EngineModel):
            def _variables(self):
                return {'W': np.arange(6, dtype=np.float32).reshape(2, 3)}
        return Stub(model_path=str(tmp_path / 'm') + '/', random_seed=5)


def test_checkpoint_write_is_atomic_and_errors_reach_the_caller(tmp_path):
    """round-2 advisor: the background checkpoint writer must not lose its exception (the reference's synchronous
    save raises), and a failed write must not leave a half-written file behind."""
    m = _StubModel.make(tmp_path)
    m._save_model()
    m._join_save()
    d = str(tmp_path / 'm')
    assert sorted(f for f in os.listdir(d) if not f.startswith('logs')) == ['model.npz', 'params.json', 'random_state.json']
    with np.load(os.path.join(d, 'model.npz')) as z:
        assert np.array_equal(z['W'], np.arange(6, dtype=np.float32).reshape(2, 3))
    before = open(os.path.join(d, 'params.json')).read()
    # make the write fail inside the writer thread

    class Bad(object):                               # a mapping whose expansion (`np.savez(path, **variables)`) fails
        def keys(self):
            raise IOError('disk full')

        def __getitem__(self, k):
            raise KeyError(k)
    m._variables = lambda: Bad()
    m._save_model()
    with pytest.raises(IOError, match='disk full'):
        m._join_save()
    m._join_save()                                   # the error is reported once
    m._variables = lambda: {'W': np.arange(6, dtype=np.float32).reshape(2, 3)}
    m._save_model()                                  # a retry after a failure writes (even an unchanged state)
    m._join_save()
    assert not [f for f in os.listdir(d) if f.endswith('.tmp')]
    with np.load(os.path.join(d, 'model.npz')) as z:     # the previous checkpoint is intact
        assert np.array_equal(z['W'], np.arange(6, dtype=np.float32).reshape(2, 3))
    assert json.loads(open(os.path.join(d, 'params.json')).read()).keys() == json.loads(before).keys()


def test_checkpoint_writer_never_blocks_the_training_loop(tmp_path, monkeypatch):
    """an epoch shorter than a checkpoint write (6 ms at 784 x 1024) must not wait for the disk: a snapshot that arrives
    while the writer is busy waits in a one-deep slot, a newer one replaces it, the newest state ends up on disk"""
    import time
    from boltzmann_machines_amd import base
    m = _StubModel.make(tmp_path)
    n = {'snap': 0, 'written': []}

    def variables():
        n['snap'] += 1
        return {'W': np.full((2, 3), n['snap'], dtype=np.float32)}
    m._variables = variables
    real = np.savez

    def slow_savez(path, **kw):
        time.sleep(0.25)
        n['written'].append(int(kw['W'][0, 0]))
        return real(path, **kw)
    monkeypatch.setattr(base.np, 'savez', slow_savez)
    t0 = time.perf_counter()
    for _ in range(4):
        m._save_model()
    assert time.perf_counter() - t0 < 0.2            # four epoch ends, no wait
    m._join_save()
    assert n['written'] == [1, 4]                    # the first snapshot and the newest one; 2 and 3 were superseded
    with np.load(os.path.join(str(tmp_path / 'm'), 'model.npz')) as z:
        assert float(z['W'][0, 0]) == 4.0
    m._save_model()                                  # the writer starts again after it went idle
    m._join_save()
    assert n['written'] == [1, 4, 5]
    m._variables = lambda: {'W': np.full((2, 3), 5, dtype=np.float32)}
    m._save_model()                                  # fit()'s closing save right after the last epoch's: same state,
    m._join_save()                                   # nothing is written again
    assert n['written'] == [1, 4, 5]
    m._variables = lambda: {'W': np.full((2, 3), 6, dtype=np.float32)}
    m._save_model()
    m._join_save()
    assert n['written'] == [1, 4, 5, 6]


def test_data_parallel_is_opt_in(monkeypatch):
    """round-2 advisor: RANK / WORLD_SIZE set by a launcher must not switch a model into data-parallel mode
    (collectives, rank-0-only checkpoints) unless BM355_DATA_PARALLEL=1 asks for it."""
    src = open(os.path.join(ROOT, 'boltzmann_machines_amd', 'base.py')).read()
    assert "BM355_DATA_PARALLEL" in src
    from boltzmann_machines_amd import parallel
    monkeypatch.setenv('RANK', '1'); monkeypatch.setenv('WORLD_SIZE', '4'); monkeypatch.setenv('LOCAL_RANK', '1')
    assert parallel.dist_env() == (1, 1, 4)
    from boltzmann_machines_amd import BernoulliRBM
    from boltzmann_machines_amd import _ffi
    rbm = BernoulliRBM(n_visible=4, n_hidden=3, model_path='/tmp/bm355_optin/')
    monkeypatch.delenv('BM355_DATA_PARALLEL', raising=False)
    if _ffi.load().bm_device_count() == 0:
        with pytest.raises(_ffi.Bm355Error):         # reaches engine creation (no GPU here) ...
            rbm._ensure_engine()
    else:
        rbm._ensure_engine()
    assert rbm._world == 1 and rbm._rank == 0 and rbm._comm is None     # ... as an independent, single-process model


def test_socket_allgather_world3():
    """the torch-free rendezvous of the direct exchange's 256-byte blobs"""
    import multiprocessing as mp
    import socket
    from boltzmann_machines_amd import parallel
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('fork')
    q = ctx.Queue()

    def run(r):
        q.put((r, parallel.socket_allgather(bytes([r]) * 256, r, 3, addr='127.0.0.1', port=port)))
    ps = [ctx.Process(target=run, args=(r,)) for r in range(3)]
    [p.start() for p in ps]
    got = dict(q.get(timeout=60) for _ in ps)
    [p.join() for p in ps]
    for r in range(3):
        assert got[r] == [bytes([0]) * 256, bytes([1]) * 256, bytes([2]) * 256]


def test_xcd_tile_map_is_a_bijection_and_cuts_modelled_traffic():
    """csrc/bm_gemm.h tile_of_block (evaluated on the host through bm_debug_tile_map): every tile of a launch is
    computed by exactly one block for any tile-matrix shape, XCD grid and column-group width; and for the two shapes
    the round-2 verdict names, the number of operand panels the 8 L2s pull in drops as intended."""
    import ctypes as C
    from boltzmann_machines_amd import _ffi
    lib = _ffi.load()

    def tmap(ti, tj, bi, bj):
        nb = ti * tj
        a, b, m = (C.c_int32 * nb)(), (C.c_int32 * nb)(), (C.c_int32 * 5)()
        _ffi.check(lib.bm_debug_tile_map(ti, tj, float(bi), float(bj), a, b, m))
        return np.array(a[:]), np.array(b[:]), list(m[:])

    rng = np.random.RandomState(0)
    shapes = [(1, 1), (1, 7), (7, 1), (3, 3), (32, 8), (25, 8), (79, 48), (157, 8), (16, 625), (5, 5), (2, 200), (9, 17)]
    shapes += [(int(rng.randint(1, 90)), int(rng.randint(1, 90))) for _ in range(40)]
    for ti, tj in shapes:
        for bi, bj in ((1e5, 2e5), (1.3e5, 1.3e5), (4e6, 1e3), (1e3, 4e6)):
            a, b, m = tmap(ti, tj, bi, bj)
            assert m[0] * m[1] == 8 and m[2] >= 1
            assert a.min() >= 0 and a.max() < ti and b.min() >= 0 and b.max() < tj, (ti, tj, m)
            assert len(set(zip(a.tolist(), b.tolist()))) == ti * tj, (ti, tj, m)

    def panels(a, b):            # operand panels each XCD touches (block b runs on XCD b % 8), summed over the XCDs
        pi = sum(len(set(a[x::8].tolist())) for x in range(8))
        pj = sum(len(set(b[x::8].tolist())) for x in range(8))
        return pi, pj
    # prop-up 784x1024x512 (32 x 8 tiles of 32 x 64): 1-D slabs read 32 + 8*8 = 96 panel units; the 4 x 2 grid 64 + 32
    a, b, m = tmap(32, 8, 784 * 32 * 4, 784 * 64 * 4)
    pi, pj = panels(a, b)
    assert (m[0], m[1]) == (4, 2) and pi * 100e3 + pj * 200e3 < 0.85 * (32 * 100e3 + 64 * 200e3)
    # 3072x5000 outer products (79 x 48 tiles of 64 x 64, K = 512): column groups that fit the L2
    a, b, m = tmap(79, 48, 512 * 64 * 4, 512 * 64 * 4)
    assert m[2] * 512 * 64 * 4 <= 2.5 * 2 ** 20 and m[2] >= 8


def test_bench_fails_loudly_without_a_gpu_and_keeps_stdout_clean():
    """bench.py's stdout is the ONE JSON line of the contract and nothing else; without a HIP device there is no line
    at all and no CPU fallback - a non-zero exit and a message on stderr."""
    import subprocess
    import sys
    from boltzmann_machines_amd import _ffi
    if _ffi.load().bm_device_count() > 0:
        pytest.skip('a GPU is visible: the bench would run')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu', '--no-others'], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    assert r.stdout == ''
    assert 'no HIP device' in r.stderr and 'no CPU fallback' in r.stderr


# ---- the pipelined epoch loop of fit() (rbm.py `_fit_epochs`): the report of epoch e is made after epoch e+1's first run of
# updates has been queued.  On the oracle engine, against the same run with the pipelining defeated (a validation set forces
# the synchronous order) - round-5 advisor.
def _fit_lines(monkeypatch, tmp_path, tag, every, X_val=None, val_every=1, boom_at=None, **extra):
    import contextlib
    import io
    import json
    from tests import oracle_engine
    oracle_engine.install(monkeypatch)
    import boltzmann_machines_amd as bm
    rs = np.random.RandomState(3)
    X = (rs.rand(60, 12) < 0.3).astype(np.float32)
    rbm = bm.BernoulliRBM(n_visible=12, n_hidden=8, batch_size=10, max_epoch=4, learning_rate=0.05, random_seed=7, verbose=True,
                          metrics_config=dict(msre=True, pll=True, l2_loss=True, train_metrics_every_iter=every,
                                              val_metrics_every_epoch=val_every, feg=False),
                          model_path=str(tmp_path / tag) + '/', **extra)
    if boom_at is not None:
        from boltzmann_machines_amd import rbm as rbm_mod
        name, at = boom_at
        orig, calls = getattr(rbm_mod.RbmEngine, name), [0]

        def boom(self, *a, **kw):
            calls[0] += 1
            if calls[0] == at:
                raise RuntimeError('boom')
            return orig(self, *a, **kw)
        monkeypatch.setattr(rbm_mod.RbmEngine, name, boom)
    buf, err = io.StringIO(), None
    with contextlib.redirect_stdout(buf):
        try:
            rbm.fit(X, X_val)
        except RuntimeError as e:
            err = str(e)
    lines = [l.strip() for l in buf.getvalue().replace('\r', '\n').split('\n') if l.strip().startswith('epoch:')]
    logs = sorted((tmp_path / tag).rglob('*.json*')) + sorted((tmp_path / tag).rglob('*.csv'))
    scal = {str(p.relative_to(tmp_path / tag)): p.read_text() for p in logs if 'logs' in str(p)}
    W = rbm.get_tf_params(scope='weights')['W'] if err is None else None
    return lines, scal, W, err


@pytest.mark.parametrize('every', [1, 3, 7, 100])
def test_pipelined_fit_reports_what_the_synchronous_loop_reports(monkeypatch, tmp_path, every):
    """same progress lines, same scalar logs, same parameters whether the epoch reports are deferred (pipelined) or made in
    place (a display dump is due every epoch: synchronous; the dump is computed on the host and leaves the RNG stream alone);
    and with a validation fetch due every second epoch (pipelined and synchronous epochs alternate) the run still reports
    every epoch once, in order"""
    lines_p, scal_p, W_p, _ = _fit_lines(monkeypatch, tmp_path, 'pipe', every)
    lines_s, scal_s, W_s, _ = _fit_lines(monkeypatch, tmp_path, 'sync', every, display_filters=4, v_shape=(3, 4))
    assert len(lines_p) == 4 and lines_p == lines_s
    assert np.array_equal(W_p, W_s)
    assert scal_p == scal_s and scal_p
    X_val = (np.random.RandomState(4).rand(20, 12) < 0.3).astype(np.float32)
    lines_m, _, _, _ = _fit_lines(monkeypatch, tmp_path, 'mixed', every, X_val=X_val, val_every=2)
    assert [l.split(';')[0] for l in lines_m] == ['epoch: %d/4' % e for e in (1, 2, 3, 4)]
    assert lines_m[0] == lines_p[0]                                   # (the first validation fetch, epoch 2, moves the RNG stream)
    assert ['val.' in l for l in lines_m] == [False, True, False, True]  # base_rbm.py:641: `val_results = {}` every epoch


@pytest.mark.parametrize('every,boom_at', [(1, ('train_step_metrics_async', 7)), (3, ('train_epoch', 3))])
def test_pipelined_fit_flushes_the_owed_report_when_the_next_epoch_aborts(monkeypatch, tmp_path, every, boom_at):
    """epoch 2 raises in its first engine call - at its first metrics fetch (every = 1: the report has just been made) or inside
    its first fused run of updates (every = 3: BEFORE the report of epoch 1 was due): either way epoch 1's progress line and
    scalar logs are made, and equal those of an undisturbed run"""
    lines_ok, scal_ok, _, _ = _fit_lines(monkeypatch, tmp_path, 'ok', every)
    lines, scal, _, err = _fit_lines(monkeypatch, tmp_path, 'boom', every, boom_at=boom_at)
    assert err == 'boom'
    assert lines == lines_ok[:1], (lines, lines_ok)
