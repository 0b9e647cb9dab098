#!/usr/bin/env python
"""One-off DBM parity stress (developer tool): random layer counts / sizes / flags, two updates each, against
the oracle under BM355_DEBUG=act_geo=<n>.  usage: BM355_DEBUG=act_geo=8 python tools/stress_parity_dbm.py [n] [seed]
UPDATES=5 in the environment: five updates per case, so that the paths that start with the third update of a handle (particle
sweeps on the second stream with their own tile, the layers' outer products on two streams) are compared too."""
import os, sys
os.environ.setdefault('OMP_NUM_THREADS', '16')          # the oracle is OpenMP (tests/conftest.py)
os.environ.setdefault('OMP_WAIT_POLICY', 'passive')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_dbm_parity_gpu import make_pair, data, assert_equal
from boltzmann_machines_amd.engine import as_device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for case in range(n):
    L = int(rng.randint(1, 4))
    V = int(rng.randint(4, 400))
    nh = [int(rng.randint(L + 1, 300)) for _ in range(L)]
    if rng.rand() < 0.6:
        V, nh = 4 * max(1, V // 4), [4 * max(1, m // 4) for m in nh]
    N, M = int(rng.randint(1, 90)), int(rng.randint(1, 90))
    kw = dict(max_mf_updates=int(rng.randint(0, 12)), mf_tol=float(10 ** rng.uniform(-7, -3)),
              l2=float(10 ** rng.uniform(-6, -2)), max_norm=float(rng.choice([np.inf, 1.0, 3.0])),
              sample_v_states=bool(rng.rand() < 0.7), sample_h_states=[bool(rng.rand() < 0.8) for _ in range(L)],
              sparsity_cost=[float(rng.choice([0., 1e-2]))] * L, sparsity_target=[0.15] * L)
    eng, twin = make_pair(V, nh, N, M, seed=50 + case, **kw)
    eng.seed(300 + case); twin.set_seed(300 + case)
    k = int(rng.randint(1, 4))
    names = ['vb', 'v'] + [b + ('' if i == 0 else '_%d' % i) for i in range(L) for b in ('W', 'hb', 'mu', 'h', 'q_means')]
    try:
        for s in range(int(os.environ.get('UPDATES', '2'))):
            X = data(N, V, case + s)
            n1, _ = eng.train_step(as_device(X), 0.03, 0.6, k)
            n2, _ = twin.train_step(X, 0.03, 0.6, k)
            assert n1 == n2, 'mean-field sweeps %d != %d' % (n1, n2)
        assert_equal(eng, twin, names)
    except AssertionError as e:
        bad += 1
        print('MISMATCH case %d V=%d nh=%r N=%d M=%d k=%d %r: %s' % (case, V, nh, N, M, k, kw, e))
    eng.close()
print('geometry %s: %d DBM cases, %d mismatches' % (os.environ.get('BM355_DEBUG', 'tuned'), n, bad))
sys.exit(1 if bad else 0)
