"""-m gpu: the chained launch (csrc/bm_chain.h) - h0 and the k Gibbs steps of a CD-k update, or the sweeps of
bm_rbm_gibbs, as ONE launch whose workgroups hand their rows to each other through the XCD's L2 - against the oracle,
bit for bit, and against the per-pass launches.  BM355_DEBUG=chain=<mode> is read when the first handle is used, so every mode runs in
its own subprocess: 2 forces the chained path wherever it is legal (also for shapes the default rule leaves alone:
fewer than 8 row blocks, several rounds per team), 0 switches it off, 1 is the default rule."""
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
from boltzmann_machines_amd.engine import as_device
from boltzmann_machines_amd._ffi import DeviceArray, UNIT_GAUSSIAN

mode = %(mode)d
chained = 0
# CD-k updates: the north-star shape (8 row blocks: one per XCD), a ragged batch (row tail, 3 row blocks), several
# rounds per team (17 row blocks), k = 3 (7 passes), Gaussian visible units, the DBM pre-training multipliers
cases = ((784, 1024, 512, 1, dict(sample_v_states=True)),
         (784, 256, 130, 2, dict(sample_v_states=True, l2=1e-4, sparsity_cost=1e-3)),
         (256, 320, 1060, 1, dict(sample_v_states=False)),
         (200, 192, 64, 3, dict(sample_v_states=True, dropout=0.8)),
         (320, 256, 96, 1, dict(v_unit=UNIT_GAUSSIAN, sample_v_states=True)),
         (208, 224, 48, 2, dict(dbm_first=True)),
         (208, 224, 48, 2, dict(dbm_last=True, sample_h_states=False)))
for V, H, B, k, kw in cases:
    eng, twin = make_pair(V, H, max_batch=B, **kw)
    eng.seed(11); twin.set_seed(11)
    gauss = kw.get('v_unit', 0) == UNIT_GAUSSIAN
    if gauss:
        sig = (0.5 + np.arange(V, dtype=np.float32) / V).astype(np.float32)
        eng.set('sigma', sig); twin.p['sigma'][...] = sig
    for s in range(3):
        X = synth_data(B, V, s, gaussian=gauss)
        eng.train_step(as_device(X), B, 0.05, 0.9, k)
        twin.train_step(X, 0.05, 0.9, k)
        assert_state_equal(eng, twin)
    # the metrics fetch and transform go through the same run of passes
    X = synth_data(B, V, 7, gaussian=gauss)
    Hd = DeviceArray((B, H))
    eng.transform(as_device(X), B, k, Hd)
    eng.sync()
    assert np.array_equal(Hd.numpy().view(np.uint32), twin.transform(X, k).view(np.uint32)), ('transform', V, H, B)
    st = eng.chain_stats()
    chained += st[0]
    assert st[2] == mode, st
    if mode == 2:
        assert st[0] >= 4, ('not chained', V, H, B, st)
    if mode == 0:
        assert st[0] == 0, st
    eng.close()

# sampling sweeps, in place in the caller's buffers: 15 sweeps = 30 passes = two launches (24 + 6)
for V, H, B, n in ((784, 1024, 512, 15), (784, 128, 64, 4), (256, 192, 700, 5)):
    eng, twin = make_pair(V, H, max_batch=B, sample_v_states=True)
    eng.seed(5); twin.set_seed(5)
    H0 = synth_data(B, H, 9)
    Hd = DeviceArray.from_numpy(H0)
    Vd = DeviceArray((B, V))
    for rep in range(2):
        eng.gibbs(Hd, Vd, B, n)
        eng.sync()
        Hc, Vc = twin.gibbs(H0, n)
        assert np.array_equal(Hd.numpy(), Hc), ('gibbs h', V, H, B, rep)
        assert np.array_equal(Vd.numpy(), Vc), ('gibbs v', V, H, B, rep)
        H0 = Hc
    chained += eng.chain_stats()[0]
    eng.close()
print('CHAIN_OK', chained)
'''


@pytest.mark.parametrize('mode', [2, 1, 0])
def test_chained_launch_bit_exact(gpu_lib, mode):
    env = dict(os.environ, BM355_DEBUG='chain=%d' % mode)
    r = subprocess.run([sys.executable, '-c', SCRIPT % dict(root=ROOT, mode=mode)], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and 'CHAIN_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    n = int(r.stdout.strip().split()[-1])
    assert (n == 0) == (mode == 0), r.stdout[-500:]


def test_soak_chained_updates_match_per_pass_launches(gpu_lib):
    """300 CD-1 updates at the north-star shape, chained (forced: the default rule chains four passes or more) against
    per-pass launches (two subprocesses): the variables
    agree bit for bit at the end (any stale read or lost hand-over would show up as a different bitmap somewhere)."""
    script = r'''
import sys, zlib
sys.path.insert(0, %(root)r)
import numpy as np
from tests.helpers import make_pair, synth_data
from boltzmann_machines_amd.engine import as_device
V, H, B = 784, 1024, 512
eng, _ = make_pair(V, H, max_batch=B, sample_v_states=True)
eng.seed(3)
X = as_device(np.concatenate([synth_data(B, V, s) for s in range(4)]))
for e in range(75):
    eng.train_epoch(X, 4 * B, B, 0.05, 0.9, 1)
eng.sync()
print('CRC', zlib.crc32(eng.get('W').tobytes()), zlib.crc32(eng.get('hb').tobytes()), eng.chain_stats()[0])
'''
    out = []
    for mode in ('2', '0'):
        r = subprocess.run([sys.executable, '-c', script % dict(root=ROOT)], env=dict(os.environ, BM355_DEBUG='chain=%s' % mode),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and 'CRC' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        out.append(r.stdout.strip().splitlines()[-1].split())
    assert out[0][1:3] == out[1][1:3], out
    assert int(out[0][3]) == 300 and int(out[1][3]) == 0, out
