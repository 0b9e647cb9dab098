"""-m gpu: the float64 RBM path (bm_rbm64_*) against the float64 oracle.  Bar: BIT-EXACT parameters,
probabilities and sample bitmaps (same canonical chains, every op in IEEE double); metrics to 1e-10."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def make_pair(V, H, B, **kw):
    from boltzmann_machines_amd.engine import RbmEngine64
    eng = RbmEngine64(V, H, max_batch=B, **kw)
    twin = orc.OracleRBM64(V, H, **kw)
    W = (orc.normal(87654321, 1337, 0, V * H).astype(np.float64) * 0.05).reshape(V, H)
    vb = (orc.uniform_d(87654321, 1338, 0, V) - 0.5) * 0.2
    hb = (orc.uniform_d(87654321, 1339, 0, H) - 0.5) * 0.2
    for name, val in (('W', W), ('vb', vb), ('hb', hb)):
        eng.set(name, val)
        twin.p[name][...] = val
    return eng, twin


def data(B, V, seed, gaussian=False):
    if gaussian:
        return orc.normal(87654321, 43 + seed, 0, B * V).astype(np.float64).reshape(B, V)
    return (orc.uniform_d(87654321, 42 + seed, 0, B * V).reshape(B, V) < 0.1307).astype(np.float64)


def dev(a):
    from boltzmann_machines_amd._ffi import DeviceArray
    return DeviceArray.from_numpy(a, np.float64)


def assert_state_equal(eng, twin):
    for n in ('W', 'vb', 'hb', 'dW', 'dvb', 'dhb', 'q_means'):
        g, c = eng.get(n), twin.p[n]
        bad = int(np.sum(g.view(np.uint64) != c.view(np.uint64)))
        assert bad == 0, '%s: %d / %d elements differ bitwise (max abs diff %.3e)' % (n, bad, g.size, np.max(np.abs(g - c)))


CASES = [
    (12, 8, 16, 1, dict(sample_v_states=True, sample_h_states=True, dropout=0.9)),        # reference test shape
    (12, 8, 5, 3, dict(sample_v_states=False, sparsity_cost=0.01)),
    (100, 52, 37, 2, dict(sample_v_states=True, l2=1e-3, sparsity_cost=1e-3, dbm_first=True)),
    (784, 128, 100, 1, dict(l2=1e-5, dbm_last=True)),
]


@pytest.mark.parametrize('V,H,B,k,kw', CASES)
def test_train_steps_bit_exact_f64(gpu_lib, V, H, B, k, kw):
    eng, twin = make_pair(V, H, B, **kw)
    eng.seed(1337); twin.set_seed(1337)
    for s in range(2):
        X = data(B, V, s)
        eng.train_step(dev(X), B, 0.05, 0.9, k)
        twin.train_step(X, 0.05, 0.9, k)
        assert_state_equal(eng, twin)
    X = data(B, V, 9)
    from boltzmann_machines_amd._ffi import DeviceArray
    Hd = DeviceArray((B, H), np.float64)
    eng.transform(dev(X), B, k, Hd)
    eng.sync()
    g, c = Hd.numpy(), twin.transform(X, k)
    assert np.array_equal(g.view(np.uint64), c.view(np.uint64))
    np.testing.assert_allclose(eng.metrics(dev(X), B, k), twin.metrics(X, k), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(eng.free_energy(dev(X), B), twin.free_energy(X), rtol=1e-12)
    eng.close()


def test_gaussian_visible_f64(gpu_lib):
    """means bit-exact without v sampling; Normal sampling goes through device log/sin/cos: 1e-12"""
    V, H, B = 48, 40, 21
    sig = np.linspace(0.5, 1.5, V)
    eng, twin = make_pair(V, H, B, v_unit=1, l2=1e-3)
    eng.set('sigma', sig); twin.p['sigma'][...] = sig
    eng.seed(3); twin.set_seed(3)
    X = data(B, V, 1, gaussian=True)
    eng.train_step(dev(X), B, 1e-3, 0.9, 2)
    twin.train_step(X, 1e-3, 0.9, 2)
    assert_state_equal(eng, twin)
    eng.close()
    eng, twin = make_pair(V, H, B, v_unit=1, sample_v_states=True)
    eng.set('sigma', sig); twin.p['sigma'][...] = sig
    eng.seed(3); twin.set_seed(3)
    eng.train_step(dev(X), B, 1e-3, 0.9, 1)
    twin.train_step(X, 1e-3, 0.9, 1)
    for n in ('W', 'vb', 'hb'):
        np.testing.assert_allclose(eng.get(n), twin.p[n], rtol=1e-11, atol=1e-13)
    eng.close()


MN_CASES = [
    (12, 8, 16, 1, 10, dict(sample_v_states=True, sample_h_states=True)),
    (40, 33, 9, 2, 25, dict(sample_v_states=True, l2=1e-3, sparsity_cost=1e-3)),
    (96, 70, 37, 1, 7, dict(sample_h_states=False, dropout=0.8)),
    (200, 130, 64, 2, 100, dict(sample_v_states=True, dbm_first=True)),     # interior tiles + K tails of the fast path
]


@pytest.mark.parametrize('V,H,B,k,M,kw', MN_CASES)
def test_multinomial_hidden_f64_bit_exact(gpu_lib, V, H, B, k, M, kw):
    """MultinomialRBM in float64 (MultinomialLayer of layers.py:54-70 as the hidden layer): parameters, softmax
    means and multinomial counts bit-identical to the float64 oracle; metrics (three independent h_hat draws per
    fetch, rbm.py:52-62) to 1e-10"""
    from boltzmann_machines_amd._ffi import DeviceArray
    eng, twin = make_pair(V, H, B, h_unit=2, n_samples=M, **kw)
    eng.seed(77); twin.set_seed(77)
    for s in range(2):
        X = data(B, V, s)
        eng.train_step(dev(X), B, 0.05, 0.9, k)
        twin.train_step(X, 0.05, 0.9, k)
        assert_state_equal(eng, twin)
    X = data(B, V, 9)
    Hd = DeviceArray((B, H), np.float64)
    eng.transform(dev(X), B, k, Hd)
    eng.sync()
    g, c = Hd.numpy(), twin.transform(X, k)
    assert np.array_equal(g.view(np.uint64), c.view(np.uint64))
    np.testing.assert_allclose(g.sum(axis=1), M, rtol=1e-12)            # means = M * softmax
    np.testing.assert_allclose(eng.metrics(dev(X), B, k), twin.metrics(X, k), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(eng.free_energy(dev(X), B), twin.free_energy(X), rtol=1e-12)
    np.testing.assert_allclose(eng.free_energy(dev(X), B), twin.free_energy(X), rtol=1e-12)   # the stream advanced alike
    eng.close()


def test_rejects_bad_multinomial_f64(gpu_lib):
    from boltzmann_machines_amd._ffi import Bm355Error
    from boltzmann_machines_amd.engine import RbmEngine64
    with pytest.raises(Bm355Error):
        RbmEngine64(8, 8, max_batch=4, h_unit=2, n_samples=0)
