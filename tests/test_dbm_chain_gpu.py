"""-m gpu: the chained DBM update (csrc/bm_dbmchain.h) - the mean-field loop (dbm.py:429-478) and the particle sweeps
(dbm.py:480-509) of one train step as workgroups of ONE launch, with the data-dependent trip count decided inside the
launch (one speculative sweep, three rotating mu buffers) - against the oracle, bit for bit, INCLUDING the executed sweep
count, and against the per-pass launches.  BM355_DEBUG=dbm_chain=<mode> is read when the first handle is used, so every mode runs in
its own subprocess: 2 forces the chained path wherever it is legal (also with fewer than 8 row blocks: teams without
rows), 0 (the default: the path is bit-exact but measured no faster, profiles/r5_dbm_chain_timeline.txt) switches it off, 1 is
the rule for where it would apply (8 row blocks, from the third update of a handle on)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from tests import test_dbm_parity_gpu as D
from boltzmann_machines_amd.engine import as_device

mode, big = %(mode)d, %(big)d
names = ['vb', 'dvb', 'v']
for sfx in ('', '_1'):
    names += [b + sfx for b in ('W', 'dW', 'hb', 'dhb', 'q_means', 'mu_means', 'mu', 'h')]
# (V, [h1, h2], rows, particles, k, updates, kwargs): early convergence (the loop ends well before max_mf_updates: the
# speculative sweep and the particle tail), the cap (tol = 0 never converges), no sweep at all (tol = 2 ends the loop at
# step 0), odd and even k (the particle buffers swap per sweep), unsampled layers, Gaussian visibles, different row
# counts of the two families, a max_mf_updates of 1 and 2 (no verdict is ever consulted)
small = [
    (256, [192, 320], 128, 128, 2, 4, dict(max_mf_updates=12, mf_tol=1e-4, l2=1e-4, max_norm=2.0,
                                            sparsity_target=[0.2, 0.1], sparsity_cost=[1e-2, 5e-3])),
    (256, [192, 320], 128, 64, 3, 3, dict(max_mf_updates=5, mf_tol=0.0)),
    (256, [192, 320], 64, 128, 1, 3, dict(max_mf_updates=7, mf_tol=2.0)),
    (320, [256, 192], 192, 192, 2, 3, dict(max_mf_updates=9, mf_tol=1e-5, sample_v_states=False, sample_h_states=[True, False])),
    (208, [224, 256], 128, 128, 2, 3, dict(max_mf_updates=6, mf_tol=1e-3, v_unit=1)),
    (256, [192, 320], 128, 128, 4, 3, dict(max_mf_updates=1, mf_tol=1e-7)),
    (256, [192, 320], 128, 128, 2, 3, dict(max_mf_updates=2, mf_tol=1e-7)),
]
full = [
    (784, [512, 1024], 512, 512, 5, 3, dict(max_mf_updates=50, mf_tol=1e-7, l2=1e-7, max_norm=6., sparsity_target=[0.2, 0.1],
                                            sparsity_cost=[1e-4, 5e-5])),
    (784, [512, 1024], 512, 512, 5, 3, dict(max_mf_updates=50, mf_tol=1e-4, l2=1e-7, max_norm=6.)),
]
total = 0
for V, nh, N, M, k, updates, kw in (full if big else small):
    eng, twin = D.make_pair(V, nh, N, M, **kw)
    eng.seed(42); twin.set_seed(42)
    trips = []
    for s in range(updates):
        X = D.data(N, V, s) if kw.get('v_unit', 0) == 0 else D.orc.normal(5, 70 + s, 0, N * V).reshape(N, V)
        n1, m1 = eng.train_step(as_device(X), 0.02, 0.5, k, want_msre=True)
        n2, m2 = twin.train_step(X, 0.02, 0.5, k, want_msre=True)
        assert n1 == n2, ('executed mean-field sweeps', s, n1, n2, V, nh, kw)
        np.testing.assert_allclose(m1, m2, rtol=1e-5)
        D.assert_equal(eng, twin, names)
        trips.append(n1)
    eng.sync()
    st = eng.chain_stats()
    total += st[0]
    assert st[2] == mode, st
    if mode == 0:
        assert st[0] == 0, st
    if mode == 2:
        assert st[0] == updates, ('not chained', V, nh, N, M, st)
    if mode == 1 and big:
        assert st[0] == updates - 2, ('default rule: chained from the third update on', st)
    print('trips', V, nh, N, M, trips, 'chained updates', st[0])
    eng.close()
print('CHAINED_UPDATES', total)
'''


def _run(mode, big):
    env = dict(os.environ, BM355_DEBUG='dbm_chain=%d' % mode)
    r = subprocess.run([sys.executable, '-c', SCRIPT % dict(root=ROOT, mode=mode, big=big)], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    print(r.stdout)
    return r.stdout


@pytest.mark.parametrize('mode', [0, 2])
def test_dbm_chain_small_shapes_bit_exact(gpu_lib, mode):
    out = _run(mode, 0)
    if mode == 2:
        assert 'CHAINED_UPDATES 0' not in out, out[-2000:]


@pytest.mark.parametrize('mode', [1, 2])
def test_dbm_chain_config3_shape_bit_exact(gpu_lib, mode):
    """784-512-1024, 512 rows + 512 particles, PCD-5, up to 50 sweeps: every team owns one row block of each family"""
    out = _run(mode, 1)
    assert 'CHAINED_UPDATES 0' not in out, out[-2000:]
