#!/usr/bin/env python
"""Developer tool: the chained DBM update (csrc/bm_dbmchain.h, forced with BM355_DEBUG=dbm_chain=2) on random 2-layer stacks
whose shapes the path accepts (64-row blocks, K >= 192, 16-byte pitches) - random tolerances (early end, cap, no sweep), sweep
caps, particle sweeps and sampling flags, three updates each - against the oracle, bit for bit incl. the executed sweep count.
usage: python tools/stress_dbm_chain.py [n] [seed]"""
import os, sys
os.environ.setdefault('BM355_DEBUG', 'dbm_chain=2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_dbm_parity_gpu import make_pair, data, assert_equal
from boltzmann_machines_amd.engine import as_device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
bad = chained = 0
for case in range(n):
    V = 4 * int(rng.randint(48, 120))
    nh = [4 * int(rng.randint(48, 100)), 4 * int(rng.randint(48, 130))]
    N, M = 64 * int(rng.randint(1, 5)), 64 * int(rng.randint(1, 5))
    kw = dict(max_mf_updates=int(rng.choice([1, 2, 3, 6, 15, 40])), mf_tol=float(rng.choice([0.0, 1e-7, 1e-5, 1e-3, 2.0])),
              l2=float(10 ** rng.uniform(-6, -2)), max_norm=float(rng.choice([np.inf, 1.0, 3.0])),
              sample_v_states=bool(rng.rand() < 0.7), sample_h_states=[bool(rng.rand() < 0.8) for _ in range(2)],
              sparsity_cost=[float(rng.choice([0., 1e-2]))] * 2, sparsity_target=[0.15] * 2)
    eng, twin = make_pair(V, nh, N, M, seed=50 + case, **kw)
    eng.seed(300 + case); twin.set_seed(300 + case)
    k = int(rng.randint(1, 6))
    names = ['vb', 'v'] + [b + s for s in ('', '_1') for b in ('W', 'dW', 'hb', 'mu', 'h', 'q_means')]
    trips = []
    try:
        for s in range(3):
            X = data(N, V, case + s)
            n1, _ = eng.train_step(as_device(X), 0.03, 0.6, k)
            n2, _ = twin.train_step(X, 0.03, 0.6, k)
            assert n1 == n2, 'mean-field sweeps %d != %d' % (n1, n2)
            trips.append(n1)
        assert_equal(eng, twin, names)
    except AssertionError as e:
        bad += 1
        print('MISMATCH case %d V=%d nh=%r N=%d M=%d k=%d %r: %s' % (case, V, nh, N, M, k, kw, e))
    chained += eng.chain_stats()[0]
    eng.close()
    print('case %d V=%d nh=%r N=%d M=%d k=%d max=%d tol=%g trips %r' % (case, V, nh, N, M, k, kw['max_mf_updates'], kw['mf_tol'], trips), flush=True)
print('%d stacks, %d chained updates, %d mismatches' % (n, chained, bad))
sys.exit(1 if bad else 0)
