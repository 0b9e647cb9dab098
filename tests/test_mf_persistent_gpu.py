"""-m gpu: the persistent mean-field kernel (csrc/bm_mf.h) against the launch-per-layer path it replaces and against
the oracle: same mu bits, same executed trip count, for several batch sizes of its shape family (784-512-1024) and
for tolerances / caps that end the loop after 0, 1, 2, few, many and max sweeps."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import test_dbm_parity_gpu as D

pytestmark = pytest.mark.gpu
V, NH = 784, [512, 1024]


def _pair(N, **kw):
    eng, twin = D.make_pair(V, NH, N, 8, **kw)
    # weights small enough for the loop to converge in a handful of sweeps at loose tolerances
    for nm, sc in (('W', 0.3), ('W_1', 0.3)):
        w = twin.p[nm] * np.float32(sc)
        eng.set(nm, w); twin.p[nm][...] = w
    return eng, twin


@pytest.mark.parametrize('N', [512, 128, 256])
@pytest.mark.parametrize('tol,cap', [(1e-7, 50), (1e-3, 50), (5e-2, 50), (0.9, 50), (1e-7, 1), (1e-7, 2), (1e-7, 3), (1e-4, 7)])
def test_persistent_equals_per_layer_and_oracle(gpu_lib, N, tol, cap):
    from boltzmann_machines_amd.engine import as_device
    eng, twin = _pair(N, max_mf_updates=cap, mf_tol=tol)
    ref, _ = _pair(N, max_mf_updates=cap, mf_tol=tol)
    eng.set_mf_persistent(True)                 # opt-in: the per-layer launches are the default
    for s in range(3):                          # the persistent mu of one call seeds the next
        X = D.data(N, V, s)
        n_p = eng.mean_field(as_device(X))
        n_r = ref.mean_field(as_device(X))
        n_o = twin.mean_field(X)
        assert n_p == n_r == n_o, (N, tol, cap, s, n_p, n_r, n_o)
        for nm in ('mu', 'mu_1'):
            g, r = eng.get(nm), ref.get(nm)
            assert np.array_equal(g.view(np.uint32), r.view(np.uint32)), (nm, s, n_p)
            assert np.array_equal(g.view(np.uint32), twin.p[nm].view(np.uint32)), (nm, s, n_p)
    eng.close(); ref.close()


def test_training_updates_with_the_persistent_kernel(gpu_lib):
    """whole DBM updates (mean-field + PCD + gradients) at the configs[3] shape: same parameters either way"""
    from boltzmann_machines_amd.engine import as_device
    kw = dict(max_mf_updates=50, mf_tol=1e-7, l2=1e-7, max_norm=6., sparsity_target=[0.2, 0.1], sparsity_cost=[1e-4, 5e-5])
    a, _ = D.make_pair(V, NH, 512, 512, **kw)
    b, _ = D.make_pair(V, NH, 512, 512, **kw)
    a.set_mf_persistent(True)
    a.seed(42); b.seed(42)
    for s in range(4):
        X = as_device(D.data(512, V, s))
        na, _ = a.train_step(X, 2e-3, 0.9, 5)
        nb, _ = b.train_step(X, 2e-3, 0.9, 5)
        assert na == nb and na > 1, (s, na, nb)
    for nm in ('W', 'W_1', 'vb', 'hb', 'hb_1', 'dW', 'dW_1', 'v', 'h', 'h_1', 'mu', 'mu_1'):
        assert np.array_equal(a.get(nm).view(np.uint32), b.get(nm).view(np.uint32)), nm
    a.close(); b.close()
