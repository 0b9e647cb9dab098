"""-m gpu: every act_kernel geometry, deterministically.  The launcher picks a geometry per shape by
measurement (bm_kernels.h launch_act), so a normal test run exercises whichever wins on that box;
here each one is forced through BM355_DEBUG=act_geo=<n> (read once per process -> one subprocess per geometry)
and must reproduce the oracle bit for bit on an RBM update, a DBM update and a mean-field pass."""
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
from tests.helpers import assert_state_equal, make_pair, synth_data
from boltzmann_machines_amd.engine import as_device, DbmEngine
from oracle import oracle as orc

# RBM: ragged tiles, K tail (careful steps), dropout, both samplers; then a 4-aligned shape (slim steps)
for V, H, B, k, kw in ((100, 52, 37, 2, dict(sample_v_states=True, l2=1e-3, sparsity_cost=1e-3, dropout=0.8)),
                       (784, 256, 64, 1, dict(sample_v_states=True, l2=1e-5)),
                       (37, 23, 19, 2, dict(dbm_first=True))):
    eng, twin = make_pair(V, H, max_batch=B, **kw)
    eng.seed(11); twin.set_seed(11)
    for s in range(2):
        X = synth_data(B, V, s)
        eng.train_step(as_device(X), B, 0.05, 0.9, k)
        twin.train_step(X, 0.05, 0.9, k)
        assert_state_equal(eng, twin)
    eng.close()

# DBM: two-segment layer inputs, mean-field residual, PCD
V, nh, N = 40, [24, 32], 16
kw = dict(n_particles=N, batch_size=N, max_mf_updates=6, mf_tol=1e-6, l2=1e-3, max_norm=2.0)
eng = DbmEngine(V, nh, **kw)
twin = orc.OracleDBM(V, nh, **kw)
W0 = (orc.normal(1, 1, 0, V * nh[0]) * np.float32(0.1)).reshape(V, nh[0])
W1 = (orc.normal(1, 2, 0, nh[0] * nh[1]) * np.float32(0.1)).reshape(nh[0], nh[1])
P0 = (orc.uniform(1, 3, 0, N * V) < 0.3).astype(np.float32).reshape(N, V)
for name, val in (('W', W0), ('W_1', W1), ('v', P0)):
    eng.set(name, val); twin.p[name][...] = val
eng.seed(7); twin.set_seed(7)
X = (orc.uniform(1, 4, 0, N * V) < 0.2).astype(np.float32).reshape(N, V)
for s in range(2):
    eng.train_step(as_device(X), 0.05, 0.5, 2)
    twin.train_step(X, 0.05, 0.5, 2)
for n in ('W', 'W_1', 'vb', 'hb', 'hb_1', 'v', 'mu', 'mu_1'):
    assert np.array_equal(eng.get(n).view(np.uint32), twin.p[n].view(np.uint32)), n
print('GEOMETRY_OK')
'''


@pytest.mark.parametrize('geo', ['8', '4', '1', '3', '108', '104', '101', '103', '208', '6', '5', '7', '9', 'g4', 'g8', 'g104', 'g108', 'g208', 'g9'])
def test_forced_geometry_bit_exact(gpu_lib, geo):
    # 8 | 4 | 1 | 3: act_kernel tile geometries with LDS-DMA staging, + 100: the same with register staging;
    # 'g4' / 'g8' (+ 100): the grad_kernel geometries (4 waves of 32 x 32, 8 waves of 32 x 16) through BM355_DEBUG=grad_geo=<n>;
    # 'g9': 8 waves with BK = 32, two workgroups per CU
    env = dict(os.environ, BM355_DEBUG=('grad_geo=' + geo[1:]) if geo.startswith('g') else ('act_geo=' + geo))
    r = subprocess.run([sys.executable, '-c', SCRIPT % dict(root=ROOT)], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and 'GEOMETRY_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
