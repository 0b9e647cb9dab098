"""-m gpu parity of the DBM path (mean-field, PCD particles, train op with sparsity and
max-norm, sample_v, reconstruction, AIS, ELBO) against the CPU oracle.
Bit-exact for everything that feeds back into state; AIS / ELBO values (fp32 sums of
softplus / entropies, reference accumulates them in arbitrary TF order) to 1e-5 relative."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def make_pair(V, nh, N, M, seed=3, **kw):
    from boltzmann_machines_amd.engine import DbmEngine
    eng = DbmEngine(V, nh, n_particles=M, batch_size=N, **kw)
    twin = orc.OracleDBM(V, nh, n_particles=M, batch_size=N, **kw)
    n = [V] + list(nh)
    for i in range(len(nh)):
        sfx = '' if i == 0 else '_%d' % i
        W = (orc.normal(87654321, seed + i, 0, n[i] * n[i + 1]) * np.float32(0.1)).reshape(n[i], n[i + 1])
        hb = (orc.uniform(87654321, seed + 10 + i, 0, n[i + 1]) - np.float32(0.5)) * np.float32(0.4)
        for nm, val in (('W' + sfx, W), ('hb' + sfx, hb)):
            eng.set(nm, val); twin.p[nm][...] = val
        Hp = (orc.uniform(87654321, seed + 20 + i, 0, M * n[i + 1]) < 0.5).astype(np.float32).reshape(M, n[i + 1])
        eng.set('h' + sfx, Hp); twin.p['h' + sfx][...] = Hp
    vb = (orc.uniform(87654321, seed + 30, 0, V) - np.float32(0.5)) * np.float32(0.4)
    eng.set('vb', vb); twin.p['vb'][...] = vb
    vp = (orc.uniform(87654321, seed + 31, 0, M * V) < 0.3).astype(np.float32).reshape(M, V)
    eng.set('v', vp); twin.p['v'][...] = vp
    return eng, twin


def data(N, V, s):
    return (orc.uniform(87654321, 99 + s, 0, N * V) < 0.2).astype(np.float32).reshape(N, V)


def assert_equal(eng, twin, names):
    for nm in names:
        g, c = eng.get(nm), twin.p[nm]
        bad = int(np.sum(g.view(np.uint32) != c.view(np.uint32)))
        assert bad == 0, '%s: %d / %d differ (max abs %.3e)' % (nm, bad, g.size, float(np.max(np.abs(g - c))))


CASES = [
    (20, [12, 16], 10, 10, dict(max_mf_updates=5, mf_tol=1e-5, l2=1e-3, max_norm=1.5,
                                sparsity_target=[0.2, 0.1], sparsity_cost=[1e-2, 5e-3])),
    (36, [24], 8, 12, dict(max_mf_updates=3, l2=1e-4)),                                   # 1 layer (RBM with PCD)
    (28, [20, 12, 8], 12, 8, dict(max_mf_updates=6, mf_tol=1e-6, max_norm=2.0)),          # 3 layers
    (784, [512, 1024], 32, 32, dict(max_mf_updates=3, mf_tol=1e-7, l2=1e-7, max_norm=6.)),  # BASELINE config[3] layer sizes
]


@pytest.mark.parametrize('V,nh,N,M,kw', CASES)
def test_train_steps_bit_exact(gpu_lib, V, nh, N, M, kw):
    from boltzmann_machines_amd.engine import as_device
    eng, twin = make_pair(V, nh, N, M, **kw)
    eng.seed(42); twin.set_seed(42)
    names = ['vb', 'dvb', 'v']
    for i in range(len(nh)):
        sfx = '' if i == 0 else '_%d' % i
        names += [b + sfx for b in ('W', 'dW', 'hb', 'dhb', 'q_means', 'mu_means', 'mu', 'h')]
    for s in range(2 if V < 500 else 1):
        X = data(N, V, s)
        n1, m1 = eng.train_step(as_device(X), 0.05, 0.5, 2, want_msre=True)
        n2, m2 = twin.train_step(X, 0.05, 0.5, 2, want_msre=True)
        assert n1 == n2
        np.testing.assert_allclose(m1, m2, rtol=1e-5)
        assert_equal(eng, twin, names)
    eng.close()


@pytest.mark.parametrize('V,nh,N,M,kw', CASES)
def test_reference_arithmetic_train_steps_bit_exact(gpu_lib, V, nh, N, M, kw):
    """bm_dbm_set_sigmoid_literal: the literal float32 tf.sigmoid (layers.py:47-48) in every pass - same trip counts, same
    bits as the oracle in ITS literal mode (orc_sigmoid_literal), and not the default arithmetic's bits"""
    from boltzmann_machines_amd.engine import as_device
    eng, twin = make_pair(V, nh, N, M, **kw)
    base, _ = make_pair(V, nh, N, M, **kw)
    eng.set_sigmoid_literal(True); twin.set_sigmoid_literal(True)
    eng.seed(42); base.seed(42); twin.set_seed(42)
    names = ['vb', 'dvb', 'v']
    for i in range(len(nh)):
        sfx = '' if i == 0 else '_%d' % i
        names += [b + sfx for b in ('W', 'dW', 'hb', 'dhb', 'q_means', 'mu_means', 'mu', 'h')]
    for s in range(2 if V < 500 else 1):
        X = data(N, V, s)
        n1, m1 = eng.train_step(as_device(X), 0.05, 0.5, 2, want_msre=True)
        n2, m2 = twin.train_step(X, 0.05, 0.5, 2, want_msre=True)
        base.train_step(as_device(X), 0.05, 0.5, 2)
        assert n1 == n2
        np.testing.assert_allclose(m1, m2, rtol=1e-5)
        assert_equal(eng, twin, names)
    mu, mu0 = eng.get('mu'), base.get('mu')
    assert not np.array_equal(mu.view(np.uint32), mu0.view(np.uint32))        # the mode does change the last bits ...
    np.testing.assert_allclose(mu, mu0, rtol=0, atol=5e-6)                     # ... and nothing else
    eng.set_sigmoid_literal(False)                                             # and it can be switched back
    twin.set_sigmoid_literal(False)
    X = data(N, V, 7)
    assert eng.train_step(as_device(X), 0.05, 0.5, 2)[0] == twin.train_step(X, 0.05, 0.5, 2)[0]
    assert_equal(eng, twin, names)
    eng.close(); base.close()


def test_reference_arithmetic_inference_and_ais_bit_exact(gpu_lib):
    """mean-field / reconstruct / sample_v bit-exact and AIS within 1e-5 of the oracle, both in the literal-sigmoid mode"""
    from boltzmann_machines_amd._ffi import DeviceArray
    from boltzmann_machines_amd.engine import as_device
    V, nh, N, M = 20, [12, 16], 10, 10
    eng, twin = make_pair(V, nh, N, M, max_mf_updates=8, mf_tol=1e-7)
    eng.set_sigmoid_literal(True); twin.set_sigmoid_literal(True)
    eng.seed(7); twin.set_seed(7)
    X = data(N, V, 5)
    top = DeviceArray((N, nh[-1]))
    assert eng.mean_field(as_device(X), out=top) == twin.mean_field(X)
    assert np.array_equal(top.numpy(), twin.p['mu_1'])
    Rd = DeviceArray((N, V))
    eng.reconstruct(as_device(X), Rd)
    eng.sync()
    assert np.array_equal(Rd.numpy(), twin.reconstruct(X))
    Vd = DeviceArray((M, V))
    eng.sample_v(3, Vd)
    assert np.array_equal(Vd.numpy(), twin.sample_v(3))
    assert_equal(eng, twin, ['v', 'h', 'h_1'])
    np.testing.assert_allclose(eng.ais(30, 8, 2, seed=5), twin.ais(30, 8, 2, 5), rtol=1e-5)
    eng.close()


def test_overlapped_particles_and_predicted_mean_field_bit_exact(gpu_lib):
    # from the third update on the PCD sweeps run on a second stream next to the mean-field, and the mean-field's
    # first group of sweeps is sized from the previous trip count (too long / too short both occur here)
    from boltzmann_machines_amd.engine import as_device
    V, nh, N, M = 28, [20, 12], 12, 8
    eng, twin = make_pair(V, nh, N, M, max_mf_updates=30, mf_tol=2e-4, l2=1e-3, max_norm=1.2)
    eng.seed(11); twin.set_seed(11)
    names = ['vb', 'dvb', 'v', 'W', 'dW', 'hb', 'dhb', 'mu', 'h', 'W_1', 'dW_1', 'hb_1', 'dhb_1', 'mu_1', 'h_1']
    counts = []
    for s in range(7):
        X = data(N, V, 40 + s) if s != 3 else np.zeros((N, V), dtype=np.float32)
        if s == 5:                              # the validation fetch in between (no update, particles advance)
            n1, m1 = eng.metrics(as_device(X), 2)
            n2, m2 = twin.metrics(X, 2)
        else:
            n1, m1 = eng.train_step(as_device(X), 0.1, 0.5, 2, want_msre=True)
            n2, m2 = twin.train_step(X, 0.1, 0.5, 2, want_msre=True)
        assert n1 == n2, (s, n1, n2)
        np.testing.assert_allclose(m1, m2, rtol=1e-5)
        assert_equal(eng, twin, names)
        counts.append(n1)
    assert len(set(counts)) > 1, counts         # the trip count did move, so the prediction was wrong at least once
    eng.close()


def test_mean_field_reconstruct_sample_v(gpu_lib):
    from boltzmann_machines_amd._ffi import DeviceArray
    from boltzmann_machines_amd.engine import as_device
    V, nh, N, M = 20, [12, 16], 10, 10
    eng, twin = make_pair(V, nh, N, M, max_mf_updates=8, mf_tol=1e-6)
    eng.seed(7); twin.set_seed(7)
    X = data(N, V, 5)
    top = DeviceArray((N, nh[-1]))
    assert eng.mean_field(as_device(X), out=top) == twin.mean_field(X)
    assert np.array_equal(top.numpy(), twin.p['mu_1'])
    Rd = DeviceArray((N, V))
    eng.reconstruct(as_device(X), Rd)
    eng.sync()
    assert np.array_equal(Rd.numpy(), twin.reconstruct(X))
    Vd = DeviceArray((M, V))
    eng.sample_v(3, Vd)
    assert np.array_equal(Vd.numpy(), twin.sample_v(3))
    assert_equal(eng, twin, ['v', 'h', 'h_1'])
    eng.close()


@pytest.mark.parametrize('k', [1, 2])
def test_ais_and_log_proba(gpu_lib, k):
    from boltzmann_machines_amd.engine import as_device
    V, nh, N, M = 20, [12, 16], 10, 10
    eng, twin = make_pair(V, nh, N, M, max_mf_updates=8, mf_tol=1e-6)
    g = eng.ais(n_betas=25, n_runs=37, k=k, seed=2222, chain0=5)
    c = twin.ais(n_betas=25, n_runs=37, k=k, seed=2222, chain0=5)
    np.testing.assert_allclose(g, c, rtol=1e-5)
    eng.seed(1); twin.set_seed(1)
    X = data(N, V, 2)
    np.testing.assert_allclose(eng.log_proba(as_device(X)), twin.log_proba(X), rtol=1e-5)
    eng.close()


def test_split_step_matches_fused(gpu_lib):
    """grad_step + apply_step (the data-parallel halves, world = 1) == fused train_step, bitwise;
    the mean-field residual goes through the injected all-reduce(max) callback."""
    from boltzmann_machines_amd import parallel
    from boltzmann_machines_amd.engine import as_device
    V, nh, N, M = 20, [12, 16], 10, 10
    kw = dict(max_mf_updates=5, mf_tol=1e-5, l2=1e-3, max_norm=1.5, sparsity_target=[0.2, 0.1], sparsity_cost=[1e-2, 5e-3])
    e1, _ = make_pair(V, nh, N, M, **kw)
    e2, _ = make_pair(V, nh, N, M, **kw)
    e1.seed(42); e2.seed(42)
    calls = []
    dp = parallel.DataParallelDBM(e2, 0, 1, lambda: None)
    e2.set_mf_allreduce(lambda x: (calls.append(x), x)[1])
    for s in range(2):
        Xd = as_device(data(N, V, s))
        n1, _ = e1.train_step(Xd, 0.05, 0.5, 2)
        n2 = dp.train_step(Xd, 0.05, 0.5, 2)
        assert n1 == n2
    assert len(calls) >= 2 and all(c >= 0 for c in calls)
    for nm in ('W', 'W_1', 'dW', 'hb', 'hb_1', 'vb', 'q_means', 'mu_means_1', 'v', 'h_1', 'mu'):
        assert np.array_equal(e1.get(nm).view(np.uint32), e2.get(nm).view(np.uint32)), nm
    g = e2.device_view('grad')
    assert g.shape[0] > 2 * (20 * 12 + 12 * 16)
    e1.close(); e2.close()


def test_gaussian_visible_pcd(gpu_lib):
    """BASELINE configs[2] in miniature: Gaussian-Bernoulli RBM trained with PCD = 1-layer DBM with a
    Gaussian visible layer (README.md:96 of the reference).  Without visible sampling everything is
    bit-exact; with Normal sampling (device logf/sincosf) parameters agree to 1e-5."""
    from boltzmann_machines_amd.engine import as_device
    V, nh, N, M = 44, [28], 12, 12
    sig = np.linspace(0.7, 1.3, V).astype(np.float32)
    for sample_v, exact in ((False, True), (True, False)):
        eng, twin = make_pair(V, nh, N, M, v_unit=1, sample_v_states=sample_v, max_mf_updates=3, l2=1e-3)
        eng.set('sigma', sig); twin.p['sigma'][...] = sig
        vp = orc.normal(87654321, 77, 0, M * V).reshape(M, V)
        eng.set('v', vp); twin.p['v'][...] = vp
        eng.seed(9); twin.set_seed(9)
        for s in range(2):
            X = orc.normal(87654321, 500 + s, 0, N * V).reshape(N, V)
            n1, _ = eng.train_step(as_device(X), 5e-3, 0.5, 3)
            n2, _ = twin.train_step(X, 5e-3, 0.5, 3)
            assert n1 == n2
        if exact:
            assert_equal(eng, twin, ['W', 'hb', 'vb', 'v', 'h', 'mu'])
        else:
            for nm in ('W', 'hb', 'vb'):
                np.testing.assert_allclose(eng.get(nm), twin.p[nm], rtol=2e-5, atol=1e-6)
        eng.close()


def test_randomised_dbm_bit_exact(gpu_lib):
    """random layer counts / sizes / flags: one DBM update + AIS sample path against the oracle"""
    from boltzmann_machines_amd.engine import as_device
    rng = np.random.RandomState(7)
    for case in range(12):
        L = int(rng.randint(1, 4))
        V = int(rng.randint(4, 90))
        nh = [int(rng.randint(L + 1, 70)) for _ in range(L)]
        if rng.rand() < 0.5:
            V, nh = 4 * max(1, V // 4), [4 * max(1, n // 4) for n in nh]
        N, M = int(rng.randint(1, 40)), int(rng.randint(1, 40))
        kw = dict(max_mf_updates=int(rng.randint(0, 7)), mf_tol=float(10 ** rng.uniform(-7, -3)),
                  l2=float(10 ** rng.uniform(-6, -2)), max_norm=float(rng.choice([np.inf, 1.0, 3.0])),
                  sample_v_states=bool(rng.rand() < 0.7), sample_h_states=[bool(rng.rand() < 0.8) for _ in range(L)],
                  sparsity_cost=[float(rng.choice([0., 1e-2]))] * L, sparsity_target=[0.15] * L)
        eng, twin = make_pair(V, nh, N, M, seed=50 + case, **kw)
        eng.seed(300 + case); twin.set_seed(300 + case)
        X = data(N, V, case)
        k = int(rng.randint(1, 4))
        n1, _ = eng.train_step(as_device(X), 0.03, 0.6, k)
        n2, _ = twin.train_step(X, 0.03, 0.6, k)
        names = ['vb', 'v'] + [b + ('' if i == 0 else '_%d' % i) for i in range(L) for b in ('W', 'hb', 'mu', 'h', 'q_means')]
        try:
            assert n1 == n2
            assert_equal(eng, twin, names)
        except AssertionError as e:
            raise AssertionError('case %d V=%d nh=%r N=%d M=%d k=%d %r: %s' % (case, V, nh, N, M, k, kw, e))
        eng.close()


def test_sample_v_zero_steps_keeps_particles(gpu_lib):
    """k = 0 (the default of DBM.sample_v): no sweep runs, v is returned unchanged (dbm.py:641-648)."""
    from boltzmann_machines_amd._ffi import DeviceArray
    V, nh, N, M = 20, [12, 16], 10, 10
    eng, twin = make_pair(V, nh, N, M)
    eng.seed(7); twin.set_seed(7)
    before = eng.get('v')
    Vd = DeviceArray((M, V))
    eng.sample_v(0, Vd)
    assert np.array_equal(Vd.numpy(), before) and np.array_equal(eng.get('v'), before)
    assert np.array_equal(twin.sample_v(0), before)
    eng.close()


def test_validation_fetch_advances_particles(gpu_lib):
    """bm_dbm_metrics = session.run([msre, n_mf_updates]) of _run_val_metrics (dbm.py:813): mean-field AND
    n_gibbs_steps PCD sweeps (control dependencies, dbm.py:521-523), no parameter update."""
    from boltzmann_machines_amd.engine import as_device
    V, nh, N, M = 20, [12, 16], 10, 10
    eng, twin = make_pair(V, nh, N, M, max_mf_updates=6, mf_tol=1e-6)
    eng.seed(5); twin.set_seed(5)
    W_before = eng.get('W')
    for s in range(2):
        X = data(N, V, s)
        g = eng.metrics(as_device(X), 2)
        c = twin.metrics(X, 2)
        assert g[0] == c[0]
        np.testing.assert_allclose(g[1], c[1], rtol=1e-5)
    assert_equal(eng, twin, ['v', 'h', 'h_1', 'mu', 'mu_1'])
    assert np.array_equal(eng.get('W'), W_before)
    eng.close()


def test_ais_is_deterministic_and_geometry_invariant(gpu_lib, monkeypatch):
    """log-weights are accumulated from per-16-column partial sums in a fixed order (no atomics):
    repeated runs are bit-identical, and so are runs with every tile geometry forced."""
    V, nh, N, M = 52, [40, 36], 8, 8
    eng, _ = make_pair(V, nh, N, M)
    a = eng.ais(n_betas=30, n_runs=300, k=1, seed=11)
    b = eng.ais(n_betas=30, n_runs=300, k=1, seed=11)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # a chain's value does not depend on which other chains run with it
    c = eng.ais(n_betas=30, n_runs=100, k=1, seed=11, chain0=150)
    assert np.array_equal(c.view(np.uint32), a[150:250].view(np.uint32))
    eng.close()


@pytest.mark.parametrize('V,nh,hu,ns,vu', [
    (24, [16, 12], [0, 2], [0, 9], 1),          # Gaussian - Bernoulli - Multinomial: the layer stack of examples/dbm_cifar.py
    (20, [12, 16, 8], [2, 0, 2], [7, 0, 5], 0),  # Multinomial first and last, Bernoulli in between
    (18, [10], [2], [6], 0),
])
def test_multinomial_layers_bit_exact(gpu_lib, V, nh, hu, ns, vu):
    """Multinomial hidden layers inside the DBM (layers.py:54-70): mean-field, PCD, train op and sample_v against the
    oracle, bit-exact (logits GEMM + one-wave-per-row softmax / categorical counts)."""
    from boltzmann_machines_amd._ffi import DeviceArray
    from boltzmann_machines_amd.engine import as_device
    N = M = 10
    kw = dict(max_mf_updates=5, mf_tol=1e-5, l2=1e-3, max_norm=1.5, sparsity_cost=[1e-2] * len(nh), sparsity_target=[0.2] * len(nh),
              h_units=hu, n_samples=ns, v_unit=vu, sample_v_states=(vu == 0))
    eng, twin = make_pair(V, nh, N, M, **kw)
    eng.seed(42); twin.set_seed(42)
    names = ['vb', 'dvb', 'v']
    for i in range(len(nh)):
        sfx = '' if i == 0 else '_%d' % i
        names += [b + sfx for b in ('W', 'dW', 'hb', 'dhb', 'q_means', 'mu_means', 'mu', 'h')]
    for s in range(2):
        X = data(N, V, s) if vu == 0 else orc.normal(87654321, 500 + s, 0, N * V).reshape(N, V)
        n1, m1 = eng.train_step(as_device(X), 0.02, 0.5, 2, want_msre=True)
        n2, m2 = twin.train_step(X, 0.02, 0.5, 2, want_msre=True)
        assert n1 == n2
        np.testing.assert_allclose(m1, m2, rtol=1e-5)
        assert_equal(eng, twin, names)
    # the counts of a multinomial layer sum to n_samples in every row
    for i, (u, n) in enumerate(zip(hu, ns)):
        if u == 2:
            h = eng.get('h' + ('' if i == 0 else '_%d' % i))
            assert np.all(h.sum(axis=1) == n) and np.all(h == np.round(h))
    Vd = DeviceArray((M, V))
    eng.sample_v(2, Vd)
    assert np.array_equal(Vd.numpy().view(np.uint32), twin.sample_v(2).view(np.uint32))
    eng.close()
