"""-m gpu: the BASELINE.json configurations at their FULL sizes (configs[2], [3], [4]; configs[0]/[1]
are cases of test_rbm_parity_gpu.py).  The OpenMP oracle needs ~10 s for one full update of each; AIS
at 20 000 chains is checked through a size-independent property (a chain's value depends only on
its own global index, so any slice of chains can be recomputed alone)."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import test_dbm_parity_gpu as D
from tests.helpers import assert_state_equal, make_pair

pytestmark = pytest.mark.gpu


def test_config2_gaussian_rbm_3072x5000_batch256(gpu_lib):
    """Gaussian-Bernoulli RBM 3072 x 5000, batch 256 (examples/dbm_cifar_naive.py:83-98): one CD-1 update
    (visible means fed back, so every value is bit-pinned) and the hidden means of 3 more chains."""
    from boltzmann_machines_amd._ffi import DeviceArray
    from boltzmann_machines_amd.engine import as_device
    V, H, B = 3072, 5000, 256
    eng, twin = make_pair(V, H, max_batch=B, w_std=0.0008, v_unit=1, sample_v_states=False, l2=0.01)
    eng.seed(5); twin.set_seed(5)
    X = orc.normal(87654321, 43, 0, B * V).reshape(B, V)
    eng.train_step(as_device(X), B, 5e-4, 0.9, 1)
    twin.train_step(X, 5e-4, 0.9, 1)
    assert_state_equal(eng, twin)
    rows = X[[0, 100, 255]]
    Hd = DeviceArray((3, H))
    eng.transform(as_device(rows), 3, 1, Hd)
    eng.sync()
    assert np.array_equal(Hd.numpy().view(np.uint32), twin.transform(rows, 1).view(np.uint32))
    eng.close()


@pytest.mark.parametrize('sample_v', [False, True])
def test_config2_grbm_pcd5_through_dbm_path(gpu_lib, sample_v):
    """BASELINE configs[2] exactly as bench.py's `Grbm` workload builds it: the 1-layer DBM path (README.md:96:
    the reference RBM class has no PCD), Gaussian visible layer, 3072 x 5000, 256 rows + 256 persistent
    particles, k = 5, max_mf_updates = 1, l2 = 0.01, W ~ N(0, 0.0008^2), lr 5e-4 (examples/dbm_cifar_naive.py:
    83-98,278-285).  One full update against the oracle: BIT-EXACT with and without visible sampling - since round 4
    the Box-Muller transform of the Normal draw is pinned operation by operation as well (csrc/bm_rng.h
    pin_log_unit / pin_sincos_2pi; round 3 allowed 1e-5 and a few flipped hidden bits here)."""
    from boltzmann_machines_amd.engine import DbmEngine, as_device
    V, H, N, K = 3072, 5000, 256, 5
    kw = dict(v_unit=1, sample_v_states=sample_v, n_particles=N, batch_size=N, max_mf_updates=1, l2=0.01)
    eng = DbmEngine(V, [H], **kw)
    twin = orc.OracleDBM(V, [H], **kw)
    W = (orc.normal(87654321, 1337, 0, V * H) * np.float32(0.0008)).reshape(V, H)
    vp = orc.normal(1, 1, 0, N * V).reshape(N, V)
    for nm, val in (('W', W), ('v', vp)):
        eng.set(nm, val); twin.p[nm][...] = val
    X = orc.normal(1, 100, 0, N * V).reshape(N, V)
    eng.seed(1); twin.set_seed(1)
    g = eng.train_step(as_device(X), 5e-4, 0.9, K, want_msre=True)
    c = twin.train_step(X, 5e-4, 0.9, K, want_msre=True)
    assert g[0] == c[0], (g, c)
    np.testing.assert_allclose(g[1], c[1], rtol=1e-5)
    D.assert_equal(eng, twin, ['W', 'dW', 'vb', 'dvb', 'hb', 'dhb', 'v', 'h', 'mu'])
    eng.close()


def test_config3_dbm_784_512_1024_batch512_pcd5(gpu_lib):
    """2-layer DBM 784-512-1024, 512 rows + 512 particles, PCD-5, up to 50 mean-field sweeps
    (examples/dbm_mnist.py:250-284): one full update, bit-exact incl. the executed sweep count."""
    from boltzmann_machines_amd.engine import as_device
    V, nh, N = 784, [512, 1024], 512
    kw = dict(max_mf_updates=50, mf_tol=1e-7, l2=1e-7, max_norm=6., sparsity_target=[0.2, 0.1],
              sparsity_cost=[1e-4, 5e-5])
    eng, twin = D.make_pair(V, nh, N, N, **kw)
    eng.seed(42); twin.set_seed(42)
    X = D.data(N, V, 1)
    g = eng.train_step(as_device(X), 2e-3, 0.9, 5, want_msre=True)
    c = twin.train_step(X, 2e-3, 0.9, 5, want_msre=True)
    assert g[0] == c[0] and g[0] > 1, (g, c)                      # same number of mean-field sweeps
    np.testing.assert_allclose(g[1], c[1], rtol=1e-5)
    D.assert_equal(eng, twin, ['W', 'W_1', 'vb', 'hb', 'hb_1', 'dW', 'dW_1', 'v', 'h', 'h_1', 'mu', 'mu_1'])
    eng.close()


def test_config3_dbm_steady_state_updates_bit_exact(gpu_lib):
    """the same stack over EIGHT updates (mean-field capped at 30 sweeps to keep the oracle in seconds): from the third update of a
    handle the particle sweeps run on a second stream beside the mean-field loop - with their own 32 x 32 tile once the previous
    trip count exceeds the number of particle sweeps - and the layers' outer products run on two streams; every
    variable and the executed sweeps stay the oracle's, bit for bit, after every update"""
    from boltzmann_machines_amd.engine import as_device
    V, nh, N = 784, [512, 1024], 512
    kw = dict(max_mf_updates=30, mf_tol=1e-7, l2=1e-7, max_norm=6., sparsity_target=[0.2, 0.1],
              sparsity_cost=[1e-4, 5e-5])
    eng, twin = D.make_pair(V, nh, N, N, **kw)
    eng.seed(42); twin.set_seed(42)
    names = ['W', 'W_1', 'vb', 'hb', 'hb_1', 'dW', 'dW_1', 'v', 'h', 'h_1', 'mu', 'mu_1', 'q_means', 'mu_means_1']
    for s in range(8):
        X = D.data(N, V, 1 + s)
        g = eng.train_step(as_device(X), 2e-3, 0.9, 5, want_msre=bool(s & 1))
        c = twin.train_step(X, 2e-3, 0.9, 5, want_msre=bool(s & 1))
        assert g[0] == c[0] and g[0] > 5, (s, g, c)
        if s & 1:
            np.testing.assert_allclose(g[1], c[1], rtol=1e-5)
        D.assert_equal(eng, twin, names)
    eng.close()


def test_config3_dbm_reference_arithmetic_batch512(gpu_lib):
    """the same update in the engine's reference arithmetic (bm_dbm_set_sigmoid_literal: tf.sigmoid as float32
    1 / (1 + exp(-x)), layers.py:47-48): bit-exact against the oracle's literal mode incl. the executed sweep count"""
    from boltzmann_machines_amd.engine import as_device
    V, nh, N = 784, [512, 1024], 512
    kw = dict(max_mf_updates=50, mf_tol=1e-7, l2=1e-7, max_norm=6., sparsity_target=[0.2, 0.1],
              sparsity_cost=[1e-4, 5e-5])
    eng, twin = D.make_pair(V, nh, N, N, **kw)
    eng.set_sigmoid_literal(True); twin.set_sigmoid_literal(True)
    eng.seed(42); twin.set_seed(42)
    X = D.data(N, V, 1)
    g = eng.train_step(as_device(X), 2e-3, 0.9, 5, want_msre=True)
    c = twin.train_step(X, 2e-3, 0.9, 5, want_msre=True)
    assert g[0] == c[0] and g[0] > 1, (g, c)
    np.testing.assert_allclose(g[1], c[1], rtol=1e-5)
    D.assert_equal(eng, twin, ['W', 'W_1', 'vb', 'hb', 'hb_1', 'dW', 'dW_1', 'v', 'h', 'h_1', 'mu', 'mu_1'])
    eng.close()


def test_config4_ais_20000_chains_slice_property(gpu_lib):
    """AIS on the 784-512-1024 DBM with 20 000 chains (12 betas here): chains [7000, 7016) and the last 8
    of the full run equal the oracle's standalone evaluation of just those chains."""
    V, nh, N, R = 784, [512, 1024], 64, 20000
    eng, twin = D.make_pair(V, nh, N, N)
    full = eng.ais(n_betas=12, n_runs=R, k=1, seed=2222, chain0=0)
    assert full.shape == (R,) and np.all(np.isfinite(full))
    for c0, n in ((7000, 16), (R - 8, 8)):
        ref = twin.ais(n_betas=12, n_runs=n, k=1, seed=2222, chain0=c0)
        np.testing.assert_allclose(full[c0:c0 + n], ref, rtol=1e-5)
    # and sharding the chains as 8 ranks would (2500 each) reproduces the single-run values exactly
    part = eng.ais(n_betas=12, n_runs=2500, k=1, seed=2222, chain0=5000)
    np.testing.assert_allclose(part, full[5000:7500], rtol=1e-6)
    eng.close()


def test_config4_ais_1000_betas_vs_oracle(gpu_lib):
    """BASELINE configs[4] at its full LENGTH: 1000 beta steps on the 784-512-1024 DBM.  64 of the 20 000
    chains (chains are independent and addressed by global index: the slice property above) against the
    oracle at rtol 1e-5; the values must also be bit-identical from run to run."""
    V, nh, N = 784, [512, 1024], 64
    eng, twin = D.make_pair(V, nh, N, N)
    g = eng.ais(n_betas=1000, n_runs=64, k=1, seed=2222, chain0=12345)
    c = twin.ais(n_betas=1000, n_runs=64, k=1, seed=2222, chain0=12345)
    np.testing.assert_allclose(g, c, rtol=1e-5)
    g2 = eng.ais(n_betas=1000, n_runs=64, k=1, seed=2222, chain0=12345)
    assert np.array_equal(g.view(np.uint32), g2.view(np.uint32))
    eng.close()


def test_config4_ais_literal_float32_accumulation(gpu_lib):
    """bm_dbm_set_ais_literal: the reference's own arithmetic (dbm.py:650-660, :708-728) - every log p*_beta(x) formed
    and added / subtracted in float32, in the graph's order - at the full 1000 betas against the oracle's literal
    twin (rtol 1e-5), deterministic, and within float32 accumulation noise of the default (double) mode."""
    V, nh, N = 784, [512, 1024], 64
    eng, twin = D.make_pair(V, nh, N, N)
    d64 = eng.ais(n_betas=1000, n_runs=64, k=1, seed=2222, chain0=12345)
    eng.set_ais_literal(True)
    g = eng.ais(n_betas=1000, n_runs=64, k=1, seed=2222, chain0=12345)
    c = twin.ais(n_betas=1000, n_runs=64, k=1, seed=2222, chain0=12345, literal=True)
    np.testing.assert_allclose(g, c, rtol=1e-5)
    g2 = eng.ais(n_betas=1000, n_runs=64, k=1, seed=2222, chain0=12345)
    assert np.array_equal(g.view(np.uint32), g2.view(np.uint32))
    assert not np.array_equal(g, d64)                      # a different accumulation ...
    np.testing.assert_allclose(g, d64, rtol=2e-5)          # ... of the same chains
    eng.set_ais_literal(False)
    assert np.array_equal(eng.ais(n_betas=1000, n_runs=64, k=1, seed=2222, chain0=12345).view(np.uint32), d64.view(np.uint32))
    eng.close()


def test_ais_on_gpu_brackets_exact_log_Z(gpu_lib):
    """Ground truth: 6-4-3 DBM, partition function summed exactly over all 2^13 states (tests/np_reference.py);
    the GPU's AIS (10 000 betas, 512 runs) estimates it within 0.02 nats / 4 standard errors."""
    from boltzmann_machines_amd.utils import log_mean_exp, log_std_exp
    from tests import np_reference as ref
    V, nh = 6, [4, 3]
    eng, twin = D.make_pair(V, nh, 4, 4, seed=11)
    for nm, scale in (('W', 8.0), ('W_1', 8.0)):       # make_pair draws N(0, 0.1^2): scale up to a non-trivial model
        w = twin.p[nm] * np.float32(scale)
        eng.set(nm, w); twin.p[nm][...] = w
    P = {k: v.astype(np.float64) for k, v in twin.p.items()}
    exact = ref.dbm_exact_log_Z(P['W'], P['W_1'], P['vb'], P['hb'], P['hb_1'])
    vals = eng.ais(n_betas=10000, n_runs=512, k=1, seed=777).astype(np.float64)
    est = log_mean_exp(vals)
    sem = np.exp(log_std_exp(vals) - est) / np.sqrt(len(vals))
    assert abs(est - exact) < max(0.02, 4 * sem), (est, exact, sem)
    eng.close()
