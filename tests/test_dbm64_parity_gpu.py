"""-m gpu: the float64 DBM path (bm_dbm64_*, csrc/bm_dbm64.hip) against the float64 oracle (oracle/bm_oracle_dbm64.c):
DBM(dtype='float64') of the reference (base/mixin.py:14-25, dbm.py:294-383 "all in model dtype").  Bar: BIT-EXACT for
everything that feeds back into state (parameters, momenta, running means, mean-field mu, particle bitmaps) and the same
executed mean-field sweeps; msre / AIS / ELBO (sums in a device order) to 1e-10."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def make_pair(V, nh, N, M, seed=3, **kw):
    from boltzmann_machines_amd.engine import DbmEngine64
    eng = DbmEngine64(V, nh, n_particles=M, batch_size=N, **kw)
    twin = orc.OracleDBM64(V, nh, n_particles=M, batch_size=N, **kw)
    n = [V] + list(nh)
    for i in range(len(nh)):
        sfx = '' if i == 0 else '_%d' % i
        W = (orc.normal(87654321, seed + i, 0, n[i] * n[i + 1]).astype(np.float64) * 0.1).reshape(n[i], n[i + 1])
        hb = (orc.uniform_d(87654321, seed + 10 + i, 0, n[i + 1]) - 0.5) * 0.4
        for nm, val in (('W' + sfx, W), ('hb' + sfx, hb)):
            eng.set(nm, val); twin.p[nm][...] = val
        Hp = (orc.uniform_d(87654321, seed + 20 + i, 0, M * n[i + 1]) < 0.5).astype(np.float64).reshape(M, n[i + 1])
        eng.set('h' + sfx, Hp); twin.p['h' + sfx][...] = Hp
    vb = (orc.uniform_d(87654321, seed + 30, 0, V) - 0.5) * 0.4
    eng.set('vb', vb); twin.p['vb'][...] = vb
    vp = (orc.uniform_d(87654321, seed + 31, 0, M * V) < 0.3).astype(np.float64).reshape(M, V)
    eng.set('v', vp); twin.p['v'][...] = vp
    return eng, twin


def data(N, V, s):
    return (orc.uniform_d(87654321, 99 + s, 0, N * V) < 0.2).astype(np.float64).reshape(N, V)


def dev(a):
    from boltzmann_machines_amd._ffi import DeviceArray
    return DeviceArray.from_numpy(np.ascontiguousarray(a, dtype=np.float64), np.float64)


def state_names(nh):
    names = ['vb', 'dvb', 'v']
    for i in range(len(nh)):
        sfx = '' if i == 0 else '_%d' % i
        names += [b + sfx for b in ('W', 'dW', 'hb', 'dhb', 'q_means', 'mu_means', 'mu', 'h')]
    return names


def assert_equal(eng, twin, names):
    for nm in names:
        g, c = eng.get(nm), twin.p[nm]
        bad = int(np.sum(g.view(np.uint64) != c.view(np.uint64)))
        assert bad == 0, '%s: %d / %d differ bitwise (max abs %.3e)' % (nm, bad, g.size, float(np.max(np.abs(g - c))))


CASES = [
    (20, [12, 16], 10, 10, dict(max_mf_updates=20, mf_tol=1e-7, l2=1e-3, max_norm=1.5,
                                sparsity_target=[0.2, 0.1], sparsity_cost=[1e-2, 5e-3])),
    (36, [24], 8, 12, dict(max_mf_updates=3, l2=1e-4)),                                   # 1 layer (RBM with PCD)
    (28, [20, 12, 8], 12, 8, dict(max_mf_updates=6, mf_tol=1e-6, max_norm=2.0)),          # 3 layers
    (100, [70, 52], 37, 21, dict(max_mf_updates=10, mf_tol=1e-9, sample_v_states=False,
                                 sample_h_states=[True, False])),                           # ragged tiles, means-only layers
    (784, [512, 1024], 32, 32, dict(max_mf_updates=3, mf_tol=1e-7, l2=1e-7, max_norm=6.)),  # BASELINE configs[3] layer sizes
]


@pytest.mark.parametrize('V,nh,N,M,kw', CASES)
def test_train_steps_bit_exact_f64(gpu_lib, V, nh, N, M, kw):
    eng, twin = make_pair(V, nh, N, M, **kw)
    eng.seed(42); twin.set_seed(42)
    for s in range(2 if V < 500 else 1):
        X = data(N, V, s)
        n1, m1 = eng.train_step(dev(X), 0.05, 0.5, 2, want_msre=True)
        n2, m2 = twin.train_step(X, 0.05, 0.5, 2, want_msre=True)
        assert n1 == n2
        np.testing.assert_allclose(m1, m2, rtol=1e-10)
        assert_equal(eng, twin, state_names(nh))
    X = data(N, V, 5)
    n1, m1 = eng.metrics(dev(X), 1)
    n2, m2 = twin.metrics(X, 1)
    assert n1 == n2
    np.testing.assert_allclose(m1, m2, rtol=1e-10)
    assert_equal(eng, twin, state_names(nh))
    eng.close()


def test_gaussian_visible_f64(gpu_lib):
    """Gaussian visible units (means only: the Normal draw goes through device log / sin / cos, like the float64 RBM)"""
    V, nh, N, M = 24, [16, 10], 9, 7
    eng, twin = make_pair(V, nh, N, M, v_unit=1, sample_v_states=False, max_mf_updates=6, mf_tol=1e-8, l2=1e-3)
    sig = np.linspace(0.6, 1.4, V)
    eng.set('sigma', sig); twin.p['sigma'][...] = sig
    eng.seed(5); twin.set_seed(5)
    X = orc.normal(87654321, 77, 0, N * V).astype(np.float64).reshape(N, V)
    for s in range(2):
        assert eng.train_step(dev(X), 1e-3, 0.9, 2)[0] == twin.train_step(X, 1e-3, 0.9, 2)[0]
        assert_equal(eng, twin, state_names(nh))
    eng.close()


def test_gaussian_visible_sampled_f64(gpu_lib):
    """sampled Gaussian visibles: the Normal draw goes through the device's double log / sin / cos, so the particles agree to
    round-off (1e-11) instead of bit for bit - as in the float64 RBM (tests/test_rbm64_parity_gpu.py); sweep counts equal"""
    V, nh, N, M = 24, [16, 10], 9, 7
    eng, twin = make_pair(V, nh, N, M, v_unit=1, sample_v_states=True, max_mf_updates=6, mf_tol=1e-8)
    sig = np.linspace(0.6, 1.4, V)
    eng.set('sigma', sig); twin.p['sigma'][...] = sig
    eng.seed(5); twin.set_seed(5)
    X = orc.normal(87654321, 78, 0, N * V).astype(np.float64).reshape(N, V)
    assert eng.train_step(dev(X), 1e-3, 0.9, 1)[0] == twin.train_step(X, 1e-3, 0.9, 1)[0]
    for nm in ('W', 'W_1', 'vb', 'hb', 'hb_1', 'v', 'mu', 'mu_1'):
        np.testing.assert_allclose(eng.get(nm), twin.p[nm], rtol=1e-11, atol=1e-13, err_msg=nm)
    for nm in ('h', 'h_1'):                                     # the hidden bitmaps are draws from means that agree to 1e-11
        assert np.mean(eng.get(nm) != twin.p[nm]) < 0.02, nm
    eng.close()


def test_inference_ais_and_elbo_f64(gpu_lib):
    from boltzmann_machines_amd._ffi import DeviceArray
    V, nh, N, M = 20, [12, 16], 10, 10
    eng, twin = make_pair(V, nh, N, M, max_mf_updates=30, mf_tol=1e-9)
    eng.seed(7); twin.set_seed(7)
    X = data(N, V, 3)
    out = DeviceArray((N, nh[-1]), np.float64)
    n1 = eng.mean_field(dev(X), out=out)
    eng.sync()
    n2 = twin.mean_field(X)
    assert n1 == n2 and n1 > 3
    assert np.array_equal(out.numpy().view(np.uint64), twin.p['mu_1'].view(np.uint64))
    R = DeviceArray((N, V), np.float64)
    eng.reconstruct(dev(X), R)
    eng.sync()
    assert np.array_equal(R.numpy().view(np.uint64), twin.reconstruct(X).view(np.uint64))
    for k in (0, 3):
        Vd = DeviceArray((M, V), np.float64)
        eng.sample_v(k, Vd)
        eng.sync()
        assert np.array_equal(Vd.numpy().view(np.uint64), twin.sample_v(k).view(np.uint64)), k
        assert_equal(eng, twin, ['v', 'h', 'h_1'])
    a1 = eng.ais(n_betas=200, n_runs=23, k=2, seed=2224, chain0=5)
    a2 = twin.ais(n_betas=200, n_runs=23, k=2, seed=2224, chain0=5)
    np.testing.assert_allclose(a1, a2, rtol=1e-11)
    # log Z_0 carries the reference's float32 log(2.) (dbm.py:731-734), in every dtype
    flat, ft = make_pair(V, nh, N, M)
    for nm in ('W', 'W_1', 'hb', 'hb_1', 'vb'):
        flat.set(nm, 0.0)
    np.testing.assert_array_equal(flat.ais(n_betas=5, n_runs=4, k=1, seed=1),
                                  np.full(4, (V + sum(nh)) * float(np.log(np.float32(2.)))))
    np.testing.assert_allclose(eng.log_proba(dev(X)), twin.log_proba(X), rtol=1e-11)
    eng.close(); flat.close()


def test_public_class_float64_runs_on_the_device(gpu_lib, tmp_path):
    """DBM(dtype='float64').fit / transform / reconstruct / sample_v / log_Z / log_proba through the public class"""
    from boltzmann_machines_amd import BernoulliRBM, DBM
    V, N = 20, 40
    X = (orc.uniform_d(87654321, 300, 0, N * V) < 0.3).astype(np.float64).reshape(N, V)
    rbm1 = BernoulliRBM(n_visible=V, n_hidden=12, max_epoch=1, batch_size=10, random_seed=1, dtype='float64', verbose=False,
                        model_path=str(tmp_path / 'r1') + '/')
    rbm1.fit(X)
    Q = rbm1.transform(X)
    rbm2 = BernoulliRBM(n_visible=12, n_hidden=16, max_epoch=1, batch_size=10, random_seed=2, dtype='float64', verbose=False,
                        model_path=str(tmp_path / 'r2') + '/')
    rbm2.fit(Q)
    dbm = DBM(rbms=[rbm1, rbm2], n_particles=10, batch_size=10, max_epoch=2, max_mf_updates=20, random_seed=3, dtype='float64',
              verbose=False, model_path=str(tmp_path / 'dbm') + '/')
    dbm.fit(X)
    w = dbm.get_tf_params(scope='weights')
    assert w['W'].dtype == np.float64 and np.all(np.isfinite(w['W']))
    assert dbm.transform(X).dtype == np.float64
    assert dbm.reconstruct(X).shape == X.shape
    assert dbm.sample_v(n_gibbs_steps=2).shape == (10, V)
    log_Z = dbm.log_Z(n_betas=50, n_runs=8, n_gibbs_steps=1)[0]
    assert np.isfinite(log_Z) and np.all(np.isfinite(dbm.log_proba(X, log_Z)))
