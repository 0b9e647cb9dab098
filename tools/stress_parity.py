#!/usr/bin/env python
"""One-off parity stress (developer tool): many random RBM shapes against the oracle under the geometry given
by BM355_DEBUG=act_geo=<n>; larger dimension range than the committed randomised test (K up to 700: several K chunks,
tails of every length).  usage: BM355_DEBUG=act_geo=8 python tools/stress_parity.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.helpers import assert_state_equal, make_pair, synth_data
from boltzmann_machines_amd.engine import as_device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for case in range(n):
    V, H = int(rng.randint(1, 700)), int(rng.randint(1, 700))
    if rng.rand() < 0.6:
        V, H = 4 * max(1, V // 4), 4 * max(1, H // 4)
    B = int(rng.randint(1, 140))
    k = int(rng.randint(1, 3))
    kw = dict(sample_v_states=bool(rng.rand() < 0.5), sample_h_states=bool(rng.rand() < 0.7),
              dbm_first=bool(rng.rand() < 0.2), dbm_last=bool(rng.rand() < 0.2), l2=float(10 ** rng.uniform(-5, -2)),
              sparsity_cost=float(rng.choice([0., 1e-3])), dropout=(None if rng.rand() < 0.6 else float(rng.uniform(0.5, 0.95))))
    if rng.rand() < 0.15:
        kw.update(h_unit=2, n_samples=int(rng.randint(1, 60)))
    eng, twin = make_pair(V, H, max_batch=B, **kw)
    eng.seed(case); twin.set_seed(case)
    try:
        for s in range(2):
            X = synth_data(B, V, s + case)
            eng.train_step(as_device(X), B, 0.05, 0.8, k)
            twin.train_step(X, 0.05, 0.8, k)
        assert_state_equal(eng, twin)
    except AssertionError as e:
        bad += 1
        print('MISMATCH case %d V=%d H=%d B=%d k=%d %r: %s' % (case, V, H, B, k, kw, e))
    eng.close()
print('geometry %s: %d cases, %d mismatches' % (os.environ.get('BM355_DEBUG', 'tuned'), n, bad))
sys.exit(1 if bad else 0)
