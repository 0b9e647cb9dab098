#!/usr/bin/env python
"""One-off parity stress of the chained launch (developer tool; csrc/bm_chain.h): random RBM shapes the chained path
accepts (16-byte operands, K >= 192), batches from one row to several rounds per team, k = 1 .. 4, updates and sampling
sweeps against the oracle, bit for bit.  usage: BM355_DEBUG=chain=2 python tools/stress_chain.py [n] [seed]"""
import os, sys
os.environ.setdefault('BM355_DEBUG', 'chain=2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.helpers import assert_state_equal, make_pair, synth_data
from boltzmann_machines_amd.engine import as_device
from boltzmann_machines_amd._ffi import DeviceArray, UNIT_GAUSSIAN
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = chained = 0
for case in range(n):
    V, H = 4 * int(rng.randint(48, 280)), 4 * int(rng.randint(48, 280))
    B = int(rng.choice([rng.randint(1, 70), rng.randint(60, 600), rng.randint(500, 1300)]))
    k = int(rng.randint(1, 5))
    gauss = rng.rand() < 0.2
    kw = dict(sample_v_states=bool(rng.rand() < 0.6), sample_h_states=bool(rng.rand() < 0.8),
              dbm_first=bool(rng.rand() < 0.15), dbm_last=bool(rng.rand() < 0.15), l2=float(10 ** rng.uniform(-5, -2)),
              sparsity_cost=float(rng.choice([0., 1e-3])), dropout=(None if rng.rand() < 0.7 else float(rng.uniform(0.5, 0.95))))
    if gauss:
        kw.update(v_unit=UNIT_GAUSSIAN)
    eng, twin = make_pair(V, H, max_batch=B, **kw)
    eng.seed(case); twin.set_seed(case)
    try:
        for s in range(2):
            X = synth_data(B, V, s + case, gaussian=gauss)
            eng.train_step(as_device(X), B, 0.05, 0.8, k)
            twin.train_step(X, 0.05, 0.8, k)
        assert_state_equal(eng, twin)
        if not gauss:
            H0 = synth_data(B, H, 5 + case)
            Hd, Vd = DeviceArray.from_numpy(H0), DeviceArray((B, V))
            ns = int(rng.randint(1, 14))
            eng.gibbs(Hd, Vd, B, ns)
            eng.sync()
            Hc, Vc = twin.gibbs(H0, ns)
            assert np.array_equal(Hd.numpy(), Hc) and np.array_equal(Vd.numpy(), Vc), 'gibbs %d sweeps' % ns
        st = eng.chain_stats()
        chained += st[0]
        assert st[0] > 0, 'not chained: %r' % (st,)
    except AssertionError as e:
        bad += 1
        print('MISMATCH case %d V=%d H=%d B=%d k=%d %r: %s' % (case, V, H, B, k, kw, e))
    eng.close()
print('chained launches %d: %d cases, %d mismatches' % (chained, n, bad))
sys.exit(1 if bad else 0)
