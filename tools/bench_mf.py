#!/usr/bin/env python
"""Time of one mean-field call (bm_dbm_mean_field) at the BASELINE configs[3] shape, persistent kernel vs one launch
per layer and sweep.  usage: python tools/bench_mf.py [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boltzmann_machines_amd.engine import DbmEngine, as_device  # noqa: E402
from boltzmann_machines_amd.utils import philox  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
V, H1, H2, N = 784, 512, 1024, 512
for persistent in (False, True, False, True):
    eng = DbmEngine(V, [H1, H2], n_particles=N, batch_size=N, max_mf_updates=50, mf_tol=1e-7)
    eng.set('W', philox.tf_random_normal((V, H1), 0.01, 1337))
    eng.set('W_1', philox.tf_random_normal((H1, H2), 0.01, 1111))
    eng.set_mf_persistent(persistent)
    X = (philox.uniform(1, 200, 0, 4 * N * V) < 0.13).astype(np.float32).reshape(4 * N, V)
    Xd = as_device(X)
    for i in range(5):
        n = eng.mean_field(Xd, row=(i % 4) * N)
    eng.sync()
    t0 = time.perf_counter()
    ns = []
    for i in range(reps):
        ns.append(eng.mean_field(Xd, row=(i % 4) * N))
    eng.sync()
    dt = (time.perf_counter() - t0) / reps
    print('persistent=%d: %.1f us per mean-field call, %.1f sweeps -> %.2f us per sweep' % (persistent, 1e6 * dt, np.mean(ns), 1e6 * dt / max(np.mean(ns), 1)))
    eng.close()
