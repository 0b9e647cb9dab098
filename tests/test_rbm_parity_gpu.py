"""-m gpu parity tests proper: HIP path (through the C-ABI) vs the CPU oracle on
identical seeds/inputs.  Bar: BIT-EXACT parameters, probabilities and sample
bitmaps (the kernels reproduce the oracle's canonical summation order); metrics
(free energy, PLL, msre) within 1e-5 relative (north_star tolerance)."""
import numpy as np
import pytest

from tests.helpers import assert_state_equal, make_pair, synth_data

pytestmark = pytest.mark.gpu

CASES = [
    # V, H, B, k, kwargs
    (12, 8, 16, 1, dict(sample_v_states=True, sample_h_states=True, dropout=0.9)),        # reference test shape
    (12, 8, 5, 3, dict(sample_v_states=False, sample_h_states=True, sparsity_cost=0.01)),
    (784, 128, 100, 1, dict(l2=1e-5)),                                                     # BASELINE configs[0]
    (100, 52, 37, 2, dict(sample_v_states=True, l2=1e-3, sparsity_cost=1e-3)),             # ragged tiles
    (37, 23, 19, 2, dict(sample_v_states=True, dbm_first=True)),                           # cols % 4 != 0 (generic RNG path)
    (64, 96, 33, 1, dict(dbm_last=True, sample_h_states=False)),
    (784, 1024, 512, 1, dict(l2=1e-5, sample_v_states=True)),                              # north-star shape
]


@pytest.mark.parametrize('V,H,B,k,kw', CASES)
def test_train_steps_bit_exact(gpu_lib, V, H, B, k, kw):
    from boltzmann_machines_amd.engine import as_device
    eng, twin = make_pair(V, H, max_batch=B, **kw)
    eng.seed(1337); twin.set_seed(1337)
    nsteps = 3 if V * H < 200000 else 2
    for s in range(nsteps):
        X = synth_data(B, V, s)
        Xd = as_device(X)
        eng.train_step(Xd, B, 0.05, 0.9, k)
        twin.train_step(X, 0.05, 0.9, k)
        assert_state_equal(eng, twin)
    eng.close()


@pytest.mark.parametrize('V,H,B,k,kw', CASES[:6])
def test_transform_bit_exact(gpu_lib, V, H, B, k, kw):
    from boltzmann_machines_amd._ffi import DeviceArray
    from boltzmann_machines_amd.engine import as_device
    eng, twin = make_pair(V, H, max_batch=B, **kw)
    eng.seed(7); twin.set_seed(7)
    X = synth_data(B, V, 3)
    Hd = DeviceArray((B, H))
    eng.transform(as_device(X), B, k, Hd)
    eng.sync()
    g = Hd.numpy()
    c = twin.transform(X, k)
    assert np.array_equal(g.view(np.uint32), c.view(np.uint32))
    eng.close()


def test_gaussian_visible(gpu_lib):
    """GaussianRBM (rbm.py:88-116): bit-exact without and WITH Normal sampling of the visibles (the Box-Muller
    transform is pinned operation by operation, csrc/bm_rng.h)."""
    from boltzmann_machines_amd.engine import as_device
    V, H, B = 48, 40, 21
    sig = np.linspace(0.5, 1.5, V).astype(np.float32)
    eng, twin = make_pair(V, H, max_batch=B, v_unit=1, sample_v_states=False, l2=1e-3)
    eng.set('sigma', sig); twin.p['sigma'][...] = sig
    eng.seed(3); twin.set_seed(3)
    for s in range(2):
        X = synth_data(B, V, s, gaussian=True)
        eng.train_step(as_device(X), B, 1e-3, 0.9, 2)
        twin.train_step(X, 1e-3, 0.9, 2)
        assert_state_equal(eng, twin)
    eng.close()
    eng, twin = make_pair(V, H, max_batch=B, v_unit=1, sample_v_states=True, l2=1e-3)
    eng.seed(3); twin.set_seed(3)
    X = synth_data(B, V, 5, gaussian=True)
    eng.set('sigma', sig); twin.p['sigma'][...] = sig
    for s in range(3):
        eng.train_step(as_device(X), B, 1e-3, 0.9, 1 + s)
        twin.train_step(X, 1e-3, 0.9, 1 + s)
        assert_state_equal(eng, twin)
    eng.close()


@pytest.mark.parametrize('V,H,B,k,kw', [CASES[0], CASES[2], CASES[3]])
def test_metrics(gpu_lib, V, H, B, k, kw):
    from boltzmann_machines_amd.engine import as_device
    eng, twin = make_pair(V, H, max_batch=B, **kw)
    eng.seed(99); twin.set_seed(99)
    X = synth_data(B, V, 1)
    g = eng.metrics(as_device(X), B, k)
    c, _ = twin.metrics(X, k)
    np.testing.assert_allclose(g, c, rtol=1e-5, atol=1e-6)
    fe = eng.free_energy(as_device(X), B)
    np.testing.assert_allclose(fe, twin.free_energy(X), rtol=1e-5)
    eng.close()


def test_split_step_matches_fused(gpu_lib):
    """grad_step + apply_step (the data-parallel halves) == fused train_step, bitwise."""
    from boltzmann_machines_amd.engine import as_device
    V, H, B = 100, 52, 37
    kw = dict(sample_v_states=True, l2=1e-3, sparsity_cost=1e-3)
    e1, twin = make_pair(V, H, max_batch=B, **kw)
    e2, _ = make_pair(V, H, max_batch=B, **kw)
    e1.seed(5); e2.seed(5)
    for s in range(2):
        Xd = as_device(synth_data(B, V, s))
        e1.train_step(Xd, B, 0.05, 0.9, 1)
        e2.grad_step(Xd, B, 1)
        e2.apply_step(B, 0.05, 0.9)
    for n in ('W', 'vb', 'hb', 'dW', 'dvb', 'dhb', 'q_means'):
        assert np.array_equal(e1.get(n).view(np.uint32), e2.get(n).view(np.uint32)), n
    e1.close(); e2.close()


def test_gibbs_sweep_bit_exact(gpu_lib):
    from boltzmann_machines_amd._ffi import DeviceArray
    V, H, B = 784, 128, 64
    eng, twin = make_pair(V, H, max_batch=B, sample_v_states=True)
    eng.seed(11); twin.set_seed(11)
    H0 = synth_data(B, H, 9)
    Hd = DeviceArray.from_numpy(H0)
    Vd = DeviceArray((B, V))
    eng.gibbs(Hd, Vd, B, 4)
    eng.sync()
    Hc, Vc = twin.gibbs(H0, 4)
    assert np.array_equal(Hd.numpy(), Hc)
    assert np.array_equal(Vd.numpy(), Vc)
    eng.close()


def test_errors(gpu_lib):
    from boltzmann_machines_amd._ffi import Bm355Error
    from boltzmann_machines_amd.engine import RbmEngine, as_device
    eng = RbmEngine(8, 4, max_batch=4)
    with pytest.raises(Bm355Error):
        eng.train_step(as_device(np.zeros((8, 8), np.float32)), 8, 0.1, 0.5, 1)   # B > max_batch
    with pytest.raises(Bm355Error):
        import ctypes as C
        from boltzmann_machines_amd._ffi import check
        buf = (C.c_float * 4)()
        check(eng.lib.bm_rbm_get_param(eng._h, b'nope', buf, 4))          # unknown variable
    with pytest.raises(Bm355Error):
        eng.train_step(as_device(np.zeros((2, 8), np.float32)), 2, 0.1, 0.5, 0)   # k < 1
    eng.close()


def test_against_committed_golden_fixture(gpu_lib):
    """HIP path vs tests/golden/rbm_12x8.npz (no oracle in the loop): the reference's test shape,
    data (RNG(1337).rand(16, 12)), dropout 0.9, both samplers on."""
    import os
    from boltzmann_machines_amd._ffi import DeviceArray
    from boltzmann_machines_amd.engine import RbmEngine, as_device
    from boltzmann_machines_amd.utils import RNG, philox
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'rbm_12x8.npz'))
    V, H = 12, 8
    X = RNG(seed=1337).rand(16, V).astype(np.float32)
    eng = RbmEngine(V, H, max_batch=10, sample_v_states=True, sample_h_states=True, dropout=0.9)
    eng.set('W', philox.tf_random_normal((V, H), 0.01, 1337))
    eng.seed(4242)
    Xd = as_device(X)
    for _ in range(3):
        eng.train_step(Xd, 10, 0.01, 0.9, 1, row=0)
        eng.train_step(Xd, 6, 0.01, 0.9, 1, row=10)
    m = eng.metrics(Xd, 10, 1)
    Hd = DeviceArray((8, H))
    eng.transform(Xd, 8, 1, Hd)
    eng.sync()
    for k in ('W', 'vb', 'hb', 'dW', 'q_means'):
        assert np.array_equal(eng.get(k).view(np.uint32), g[k].view(np.uint32)), k
    np.testing.assert_allclose(m, g['metrics'], rtol=1e-5, atol=1e-6)
    assert np.array_equal(Hd.numpy(), g['transform'])
    eng.close()


def test_multinomial_against_committed_golden_fixture(gpu_lib):
    """HIP path vs tests/golden/mrbm_12x8.npz (no oracle in the loop)"""
    import os
    from boltzmann_machines_amd._ffi import DeviceArray
    from boltzmann_machines_amd.engine import RbmEngine, as_device
    from boltzmann_machines_amd.utils import RNG, philox
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mrbm_12x8.npz'))
    V, H = 12, 8
    X = RNG(seed=1337).rand(16, V).astype(np.float32)
    eng = RbmEngine(V, H, max_batch=10, sample_v_states=True, sample_h_states=True, h_unit=2, n_samples=10)
    eng.set('W', philox.tf_random_normal((V, H), 0.01, 1337))
    eng.seed(4242)
    Xd = as_device(X)
    for _ in range(3):
        eng.train_step(Xd, 10, 0.01, 0.9, 1, row=0)
        eng.train_step(Xd, 6, 0.01, 0.9, 1, row=10)
    for k in ('W', 'vb', 'hb', 'dW', 'q_means'):
        assert np.array_equal(eng.get(k).view(np.uint32), g[k].view(np.uint32)), k
    Hd = DeviceArray((8, H))
    eng.transform(Xd, 8, 1, Hd)
    eng.sync()
    assert np.array_equal(Hd.numpy(), g['transform'])
    eng.close()


def _random_cases(n, seed):
    rng = np.random.RandomState(seed)
    for _ in range(n):
        V, H = int(rng.randint(1, 200)), int(rng.randint(1, 200))
        if rng.rand() < 0.5:                       # half of the cases on the 16-byte fast path
            V, H = 4 * max(1, V // 4), 4 * max(1, H // 4)
        B = int(rng.randint(1, 70))
        k = int(rng.randint(1, 4))
        kw = dict(sample_v_states=bool(rng.rand() < 0.5), sample_h_states=bool(rng.rand() < 0.7),
                  dbm_first=bool(rng.rand() < 0.2), dbm_last=bool(rng.rand() < 0.2),
                  l2=float(10 ** rng.uniform(-5, -2)), sparsity_cost=float(rng.choice([0., 1e-3])),
                  dropout=(None if rng.rand() < 0.6 else float(rng.uniform(0.5, 0.95))))
        yield V, H, B, k, kw


def test_randomised_shapes_and_flags_bit_exact(gpu_lib):
    """40 random (V, H, B, k, flags) draws incl. 1-wide layers, B = 1, ragged tiles, both load paths:
    parameters after two updates and the transform output are bit-identical to the oracle."""
    from boltzmann_machines_amd._ffi import DeviceArray
    from boltzmann_machines_amd.engine import as_device
    for case, (V, H, B, k, kw) in enumerate(_random_cases(40, 2024)):
        eng, twin = make_pair(V, H, max_batch=B, **kw)
        eng.seed(1000 + case); twin.set_seed(1000 + case)
        for s in range(2):
            X = synth_data(B, V, s + case)
            eng.train_step(as_device(X), B, 0.05, 0.8, k)
            twin.train_step(X, 0.05, 0.8, k)
        try:
            assert_state_equal(eng, twin)
            Hd = DeviceArray((B, H))
            Xt = synth_data(B, V, 77 + case)
            eng.transform(as_device(Xt), B, k, Hd)
            eng.sync()
            assert np.array_equal(Hd.numpy().view(np.uint32), twin.transform(Xt, k).view(np.uint32))
        except AssertionError as e:
            raise AssertionError('case %d V=%d H=%d B=%d k=%d %r: %s' % (case, V, H, B, k, kw, e))
        eng.close()


def test_short_last_batch_and_epoch_driver(gpu_lib):
    """bm_rbm_train_epoch == the per-batch loop of base_rbm.py:549-571 incl. the short last batch (utils.py:37)."""
    from boltzmann_machines_amd.engine import as_device
    V, H, N, bs = 40, 24, 53, 16
    eng, twin = make_pair(V, H, max_batch=bs, sample_v_states=True)
    eng.seed(5); twin.set_seed(5)
    X = synth_data(N, V, 3)
    eng.train_epoch(as_device(X), N, bs, 0.05, 0.9, 2)
    for s in range(0, N, bs):
        twin.train_step(X[s:s + bs], 0.05, 0.9, 2)
    assert_state_equal(eng, twin)
    eng.close()


# ---- MultinomialRBM (reference rbm/rbm.py:25-65, layers.py:54-70)
MN_CASES = [
    # V, H (= states of the multinomial unit), B, k, n_samples, kwargs
    (12, 8, 16, 1, 10, dict(sample_v_states=True)),                  # reference test shape
    (100, 52, 37, 2, 100, dict(sample_v_states=True, l2=1e-3)),      # ragged tiles, default n_samples
    (64, 300, 20, 1, 7, dict(sample_h_states=False)),                # means fed back instead of counts
    (784, 1024, 128, 1, 100, dict(l2=1e-5, sample_v_states=True)),
]


@pytest.mark.parametrize('V,H,B,k,M,kw', MN_CASES)
def test_multinomial_train_steps_bit_exact(gpu_lib, V, H, B, k, M, kw):
    """softmax means, multinomial counts and the CD-k update built on them are bit-exact"""
    from boltzmann_machines_amd._ffi import DeviceArray
    from boltzmann_machines_amd.engine import as_device
    eng, twin = make_pair(V, H, max_batch=B, w_std=0.1, h_unit=2, n_samples=M, **kw)
    eng.seed(99); twin.set_seed(99)
    for s in range(2):
        X = synth_data(B, V, s)
        eng.train_step(as_device(X), B, 0.01, 0.9, k)
        twin.train_step(X, 0.01, 0.9, k)
        assert_state_equal(eng, twin)
    # hidden means after the chain (transform) and the sampling sweep
    X = synth_data(B, V, 7)
    Hd = DeviceArray((B, H))
    eng.transform(as_device(X), B, k, Hd)
    eng.sync()
    g, c = Hd.numpy(), twin.transform(X, k)
    assert np.array_equal(g.view(np.uint32), c.view(np.uint32))
    np.testing.assert_allclose(g.sum(axis=1), M, rtol=1e-5)
    Hs = twin.work['hs'].copy()
    if kw.get('sample_h_states', True):
        assert np.all(Hs == np.round(Hs)) and np.all(Hs.sum(axis=1) == M)      # counts of M draws
    Hg, Vg = DeviceArray.from_numpy(Hs), DeviceArray((B, V))
    eng.gibbs(Hg, Vg, B, 2)
    eng.sync()
    Hc, Vc = twin.gibbs(Hs, 2)
    assert np.array_equal(Hg.numpy(), Hc) and np.array_equal(Vg.numpy(), Vc)
    eng.close()


def test_multinomial_metrics_and_free_energy(gpu_lib):
    """rbm.py:52-62: free energy with a random h_hat per evaluation + the lgamma constant; PLL from
    two more draws.  Tolerance-checked (double vs fp32 accumulation)."""
    from boltzmann_machines_amd.engine import as_device
    V, H, B, M = 40, 24, 32, 50
    eng, twin = make_pair(V, H, max_batch=B, w_std=0.1, h_unit=2, n_samples=M, sample_v_states=True)
    eng.seed(5); twin.set_seed(5)
    X = synth_data(B, V, 1)
    g = eng.metrics(as_device(X), B, 1)
    c, _ = twin.metrics(X, 1)
    np.testing.assert_allclose(g, c, rtol=2e-5, atol=1e-5)
    fg = eng.free_energy(as_device(X), B)
    fc = twin.free_energy(X)
    np.testing.assert_allclose(fg, fc, rtol=2e-5)
    # the h_hat draw advanced the stream on both sides: the next step still matches
    eng.train_step(as_device(X), B, 0.01, 0.9, 1)
    twin.train_step(X, 0.01, 0.9, 1)
    assert_state_equal(eng, twin)
    eng.close()


def test_multinomial_rejects_bad_config(gpu_lib):
    from boltzmann_machines_amd._ffi import Bm355Error
    from boltzmann_machines_amd.engine import RbmEngine
    with pytest.raises(Bm355Error):
        RbmEngine(8, 8, max_batch=4, h_unit=2, n_samples=0)
    with pytest.raises(Bm355Error):
        RbmEngine(8, 9000, max_batch=4, h_unit=2, n_samples=10)
    with pytest.raises(Bm355Error):
        RbmEngine(8, 8, max_batch=4, h_unit=7)


def test_long_run_stays_bit_exact(gpu_lib):
    """300 consecutive CD-1 updates (the launch tuner changes the act_kernel geometry during the first
    launches of each shape; lr/momentum/k schedules change mid-run): parameters still bit-identical."""
    from boltzmann_machines_amd.engine import as_device
    V, H, B = 200, 96, 48
    eng, twin = make_pair(V, H, max_batch=B, sample_v_states=True, l2=1e-4, sparsity_cost=1e-3, dropout=0.9)
    eng.seed(2024); twin.set_seed(2024)
    X = synth_data(4 * B, V, 0)
    Xd = as_device(X)
    for step in range(300):
        lr, mom, k = (0.05, 0.5, 1) if step < 150 else (0.01, 0.9, 2)
        row = (step % 4) * B
        eng.train_step(Xd, B, lr, mom, k, row=row)
        twin.train_step(X[row:row + B], lr, mom, k)
        if step in (0, 8, 9, 10, 149, 150):
            assert_state_equal(eng, twin)
    assert_state_equal(eng, twin)
    assert np.all(np.isfinite(eng.get('W')))
    eng.close()


def test_train_epoch_equals_the_loop_over_train_step(gpu_lib):
    """bm_rbm_train_epoch == the caller's own loop over bm_rbm_train_step (same launches, same RNG call counters),
    with a ragged last batch and single steps in between"""
    from boltzmann_machines_amd.engine import as_device
    V, H, B = 96, 64, 16
    eng, twin = make_pair(V, H, max_batch=B, sample_v_states=True, l2=1e-4)
    X = synth_data(5 * B + 7, V, 5)
    Xd = as_device(X)
    eng.seed(21); twin.set_seed(21)
    for rep in range(3):
        eng.train_epoch(Xd, len(X), B, 0.05, 0.5, 1)
        for s in range(0, len(X), B):
            twin.train_step(X[s:s + B], 0.05, 0.5, 1)
        eng.train_step(Xd, B, 0.02, 0.9, 2, row=B)
        twin.train_step(X[B:2 * B], 0.02, 0.9, 2)
    assert_state_equal(eng, twin)
    eng.close()


def test_deferred_metrics_equal_the_synchronous_fetch(gpu_lib):
    """bm_rbm_train_step_metrics_async + bm_rbm_collect_metrics: the same values, in order, as the synchronous fetch
    (base_rbm.py:554-571), the same parameters afterwards; updates in between do not disturb pending fetches"""
    from boltzmann_machines_amd.engine import as_device
    V, H, B = 96, 64, 16
    kw = dict(max_batch=B, sample_v_states=True, l2=1e-4)
    a, twin = make_pair(V, H, **kw)
    b, _ = make_pair(V, H, **kw)
    X = synth_data(6 * B, V, 9)
    Xd = as_device(X)
    for e in (a, b):
        e.seed(33)
    want = []
    for i in range(6):
        row = i * B
        if i % 2:
            want.append(a.train_step_metrics(Xd, B if i != 5 else B - 3, 0.05, 0.5, 1, row=row))
            b.train_step_metrics_async(Xd, B if i != 5 else B - 3, 0.05, 0.5, 1, row=row)
        else:
            a.train_step(Xd, B, 0.05, 0.5, 1, row=row)
            b.train_step(Xd, B, 0.05, 0.5, 1, row=row)
    got = b.collect_metrics()
    assert got.shape == (3, 4)
    assert np.array_equal(got, np.array(want, dtype=np.float32))
    assert b.collect_metrics().shape == (0, 4)
    for name in ('W', 'vb', 'hb', 'dW', 'dvb', 'dhb'):
        assert np.array_equal(a.get(name), b.get(name)), name
    a.close(); b.close()


def test_every_writer_of_W_keeps_the_transposed_weights_in_step(gpu_lib):
    """The fused update writes W and its transpose (the x-major prop-up operand, bm_rbm.hip `Wt`); every OTHER writer of
    W - set_param, the split step's apply, set_from_device - must send the prop-up back to W itself until the next
    fused update.  A stale transpose would show up as different hidden bitmaps: the mixed sequence below stays
    bit-identical to the oracle (and to a handle that never uses the transpose: gradient steps only)."""
    from boltzmann_machines_amd._ffi import DeviceArray
    from boltzmann_machines_amd.engine import as_device
    from oracle import oracle as orc
    V, H, B = 256, 192, 64
    kw = dict(sample_v_states=True, l2=1e-4)
    eng, twin = make_pair(V, H, max_batch=B, **kw)
    eng.seed(21); twin.set_seed(21)
    W2 = (orc.normal(5, 6, 0, V * H) * np.float32(0.02)).reshape(V, H)

    def fused(s):
        X = synth_data(B, V, s)
        eng.train_step(as_device(X), B, 0.05, 0.9, 2)
        twin.train_step(X, 0.05, 0.9, 2)

    def split(s):
        X = synth_data(B, V, s)
        eng.grad_step(as_device(X), B, 2)
        eng.apply_step(B, 0.05, 0.9)
        raw = twin.raw_grads(X, 2)
        twin.apply(raw, B, 0.05, 0.9)

    fused(0); fused(1)                  # transpose valid from the first fused update on
    split(2)                            # apply_step rewrites W without the transpose
    fused(3)
    assert_state_equal(eng, twin)
    eng.set('W', W2); twin.p['W'][...] = W2
    fused(4); fused(5)
    assert_state_equal(eng, twin)
    X = synth_data(B, V, 9)
    Hd = DeviceArray((B, H))
    eng.transform(as_device(X), B, 1, Hd)
    eng.sync()
    assert np.array_equal(Hd.numpy().view(np.uint32), twin.transform(X, 1).view(np.uint32))
    eng.close()
