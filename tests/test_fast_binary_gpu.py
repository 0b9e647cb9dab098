"""-m gpu: the opt-in fast-binary mode (csrc/bm_bf3.h; SURVEY 7 hard parts 2 and 4): contractions with a {0,1}
operand run as exact-product bf16 x 3 on the bf16 matrix cores.  Products are exact, only the order of the fp32
additions differs from the canonical chain, so the bar is a TOLERANCE, with the number of draws that land on the
other side of `u < p` counted and bounded:
  * one sampling sweep of the 784 x 1024 RBM: bitmaps equal the default path's (= the oracle's) except for ties;
  * AIS log-weights of short runs equal the default path's to 1e-5 for every chain whose bitmaps did not fork;
  * the AIS estimate of a model with an exactly enumerable partition function brackets the exact log Z;
  * non-bitmap input is refused; odd shapes (K tails, tiny layers) go through the zero-padded planes."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import test_dbm_parity_gpu as D

pytestmark = pytest.mark.gpu


def _rbm(V, H, B, fast, seed=1337):
    from boltzmann_machines_amd.engine import RbmEngine
    eng = RbmEngine(V, H, max_batch=B, sample_v_states=True, sample_h_states=True)
    W = (orc.normal(87654321, seed, 0, V * H) * np.float32(0.05)).reshape(V, H)
    eng.set('W', W)
    eng.set('vb', (orc.uniform(1, 2, 0, V) - np.float32(0.5)) * np.float32(0.3))
    eng.set('hb', (orc.uniform(1, 3, 0, H) - np.float32(0.5)) * np.float32(0.3))
    eng.seed(7)
    eng.set_fast_binary(fast, everywhere=True)
    return eng


@pytest.mark.parametrize('V,H,B', [(784, 1024, 512), (100, 52, 24), (13, 70, 5), (64, 64, 64)])
def test_gibbs_sweep_bitmaps_equal_up_to_ties(gpu_lib, V, H, B):
    from boltzmann_machines_amd._ffi import DeviceArray
    h0 = (orc.uniform(5, 6, 0, B * H) < 0.5).astype(np.float32).reshape(B, H)
    out = {}
    for fast in (False, True):
        eng = _rbm(V, H, B, fast)
        Hd, Vd = DeviceArray.from_numpy(h0), DeviceArray((B, V))
        eng.gibbs(Hd, Vd, B, 1)
        eng.sync()
        out[fast] = (Vd.numpy().copy(), Hd.numpy().copy())
        eng.close()
    dv = int(np.sum(out[True][0] != out[False][0]))
    # the hidden draw follows the visible one: compare it only on rows whose visible bitmap is unchanged
    same_rows = np.all(out[True][0] == out[False][0], axis=1)
    dh = int(np.sum(out[True][1][same_rows] != out[False][1][same_rows]))
    print('fast-binary sweep %dx%d batch %d: %d / %d visible and %d / %d hidden draws differ (ties at fp32 round-off)'
          % (V, H, B, dv, B * V, dh, int(same_rows.sum()) * H))
    assert dv <= max(2, B * V // 100000) and dh <= max(2, B * H // 100000)
    assert set(np.unique(out[True][0])) <= {0.0, 1.0}


def test_gibbs_many_sweeps_statistics(gpu_lib):
    """after the first tie the two chains are different samples of the same distribution: compare moments"""
    from boltzmann_machines_amd._ffi import DeviceArray
    V, H, B = 200, 120, 256
    h0 = (orc.uniform(5, 6, 0, B * H) < 0.5).astype(np.float32).reshape(B, H)
    m = {}
    for fast in (False, True):
        eng = _rbm(V, H, B, fast)
        Hd, Vd = DeviceArray.from_numpy(h0), DeviceArray((B, V))
        eng.gibbs(Hd, Vd, B, 50)
        eng.sync()
        m[fast] = (Vd.numpy().mean(axis=0), Hd.numpy().mean(axis=0))
        eng.close()
    assert np.abs(m[True][0] - m[False][0]).max() < 0.2 and abs(m[True][0].mean() - m[False][0].mean()) < 0.01
    assert abs(m[True][1].mean() - m[False][1].mean()) < 0.01


def test_non_bitmap_input_is_refused(gpu_lib):
    from boltzmann_machines_amd import _ffi
    from boltzmann_machines_amd._ffi import DeviceArray
    eng = _rbm(32, 16, 4, True)
    Hd, Vd = DeviceArray.from_numpy(np.full((4, 16), 0.5, dtype=np.float32)), DeviceArray((4, 32))
    eng.gibbs(Hd, Vd, 4, 1)
    with pytest.raises(_ffi.Bm355Error, match='bitmap'):
        eng.sync()
    eng.close()


@pytest.mark.parametrize('V,nh,R', [(784, [512, 1024], 256), (20, [12, 16], 40), (70, [33, 9], 17),
                                     (784, [512, 1024], 2304), (100, [64, 48], 2100)])
def test_ais_short_runs_match_the_default_path(gpu_lib, V, nh, R):
    eng, _ = D.make_pair(V, nh, 8, 8)
    ref = eng.ais(n_betas=4, n_runs=R, k=1, seed=2222)
    eng.set_fast_binary(True, everywhere=True)
    fast = eng.ais(n_betas=4, n_runs=R, k=1, seed=2222)
    fast2 = eng.ais(n_betas=4, n_runs=R, k=1, seed=2222)
    eng.set_fast_binary(False)
    again = eng.ais(n_betas=4, n_runs=R, k=1, seed=2222)
    assert np.array_equal(ref.view(np.uint32), again.view(np.uint32))        # the default path is untouched
    assert np.array_equal(fast.view(np.uint32), fast2.view(np.uint32))       # and the fast one is deterministic
    close = np.isclose(fast, ref, rtol=1e-5, atol=1e-4)
    forked = int((~close).sum())
    print('fast-binary AIS %s, %d chains x 4 betas: %d chains forked on a tie; the others agree to 1e-5' % ([V] + nh, R, forked))
    assert forked <= max(1, R // 50)
    eng.close()


def test_ais_fast_brackets_exact_log_Z(gpu_lib):
    """ground truth (tests/np_reference.dbm_exact_log_Z): 6-4-3 DBM, 2^13 states summed exactly"""
    from boltzmann_machines_amd.utils import log_mean_exp, log_std_exp
    from tests import np_reference as ref
    eng, twin = D.make_pair(6, [4, 3], 4, 4, seed=11)
    for nm in ('W', 'W_1'):
        w = twin.p[nm] * np.float32(8.0)
        eng.set(nm, w); twin.p[nm][...] = w
    P = {k: v.astype(np.float64) for k, v in twin.p.items()}
    exact = ref.dbm_exact_log_Z(P['W'], P['W_1'], P['vb'], P['hb'], P['hb_1'])
    eng.set_fast_binary(True, everywhere=True)
    vals = eng.ais(n_betas=10000, n_runs=512, k=1, seed=777).astype(np.float64)
    est = log_mean_exp(vals)
    sem = np.exp(log_std_exp(vals) - est) / np.sqrt(len(vals))
    assert abs(est - exact) < max(0.02, 4 * sem), (est, exact, sem)
    eng.close()


def test_ais_estimate_consistent_at_config4_shape(gpu_lib):
    """784-512-1024, 2048 chains x 100 betas: the two modes estimate the same log Z (different samples after the first
    tie, so the comparison is statistical)"""
    from boltzmann_machines_amd.utils import log_mean_exp, log_std_exp
    eng, _ = D.make_pair(784, [512, 1024], 8, 8)
    for nm in ('W', 'W_1'):
        eng.set(nm, eng.get(nm) * np.float32(0.3))
    est = {}
    for fast in (False, True):
        eng.set_fast_binary(fast, everywhere=True)
        v = eng.ais(n_betas=100, n_runs=2048, k=1, seed=99).astype(np.float64)
        e = log_mean_exp(v)
        est[fast] = (e, np.exp(log_std_exp(v) - e) / np.sqrt(len(v)))
    d = abs(est[True][0] - est[False][0])
    assert d < 5 * np.hypot(est[True][1], est[False][1]) + 1e-3 * abs(est[False][0]), est
    eng.close()


PCD_CASES = [
    (784, [512, 1024], 32, 64, dict(max_mf_updates=3, mf_tol=1e-7, l2=1e-7, max_norm=6.), 5),   # BASELINE config[3] layers
    (20, [12, 16], 10, 10, dict(max_mf_updates=5, mf_tol=1e-5, l2=1e-3), 3),
    (36, [24], 8, 12, dict(max_mf_updates=3, l2=1e-4), 4),                                        # 1 layer
    (28, [20, 12, 8], 12, 8, dict(max_mf_updates=4, max_norm=2.0), 2),                            # 3 layers
    (70, [33, 9], 17, 5, dict(max_mf_updates=2), 1),                                              # k = 1, K tails
    (44, [28], 12, 12, dict(v_unit=1, sample_v_states=False, max_mf_updates=3, l2=1e-3), 3),      # Gaussian means
    (3072, [5000], 16, 32, dict(v_unit=1, sample_v_states=True, max_mf_updates=1, l2=0.01), 5),   # BASELINE config[2] shape
]


@pytest.mark.parametrize('V,nh,N,M,kw,k', PCD_CASES)
def test_pcd_sweeps_match_the_default_path_up_to_ties(gpu_lib, V, nh, N, M, kw, k):
    """fast-binary in the PCD particle sweeps of a train step (bm_dbm.hip particles_update): every contraction whose
    state operand is a bitmap SAMPLED EARLIER IN THE SAME CALL runs from its bf16 shadow; the particles a call starts
    from (possibly real valued) and Gaussian visibles are read in fp32.  Same chains as the default path except where
    a draw lands within round-off of its probability; parameters agree to the weight of the forked chains."""
    from boltzmann_machines_amd.engine import as_device
    gauss = kw.get('v_unit', 0) == 1
    scale = 0.1 if V < 1000 else 0.008
    out = {}
    for mode in ('default', 'fast', 'fast2'):
        eng, twin = D.make_pair(V, nh, N, M, **kw)
        if V >= 1000:
            eng.set('W', twin.p['W'] * np.float32(scale / 0.1))
        if gauss:
            eng.set('v', orc.normal(87654321, 77, 0, M * V).reshape(M, V))
        eng.seed(42)
        eng.set_fast_binary(mode != 'default', everywhere=True)
        X = orc.normal(87654321, 500, 0, N * V).reshape(N, V) if gauss else D.data(N, V, 0)
        lr = 5e-3 if gauss else 0.05
        for s in range(2):
            eng.train_step(as_device(X), lr, 0.5, k)
        names = ['v', 'vb', 'W', 'hb', 'h'] + [b + '_%d' % i for i in range(1, len(nh)) for b in ('W', 'hb', 'h')]
        out[mode] = {nm: eng.get(nm).copy() for nm in names}
        eng.close()
    for nm, a in out['fast'].items():                       # the fast mode is deterministic
        assert np.array_equal(a.view(np.uint32), out['fast2'][nm].view(np.uint32)), nm
    ref, fast = out['default'], out['fast']
    forked = np.zeros(M, bool)
    for nm in ref:
        if nm == 'h' or nm.startswith('h_'):
            assert set(np.unique(fast[nm])) <= {0.0, 1.0}
            forked |= np.any(ref[nm] != fast[nm], axis=1)
    if gauss:
        forked |= np.any(~np.isclose(ref['v'], fast['v'], rtol=1e-4, atol=1e-4), axis=1)
    else:
        forked |= np.any(ref['v'] != fast['v'], axis=1)
    nf = int(forked.sum())
    print('fast-binary PCD-%d %s, %d particles, 2 updates: %d chains forked on a tie' % (k, [V] + nh, M, nf))
    assert nf <= max(1, M // 16)
    # a forked chain moves the negative statistics by at most lr / M per element per update
    atol = 1e-6 + 2.5 * (5e-3 if gauss else 0.05) * nf / M
    for nm in ref:
        if nm[0] in 'Wvh' and nm not in ('v', 'h') and not nm.startswith('h_'):
            np.testing.assert_allclose(fast[nm], ref[nm], rtol=2e-5, atol=atol, err_msg=nm)
