#!/usr/bin/env python
"""DBM 784-512-1024 updates only (BASELINE configs[3] shape), for rocprofv3 kernel traces and A/B runs of the chained update
(BM355_DEBUG=dbm_chain=0|1, csrc/bm_dbmchain.h):  python tools/dbm_update_probe.py [rows=512] [updates=40] [mf_tol=1e-7] [k=5]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boltzmann_machines_amd.engine import DbmEngine, as_device
from boltzmann_machines_amd.utils import philox

V, H1, H2 = 784, 512, 1024
N = M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-7
k = int(sys.argv[4]) if len(sys.argv) > 4 else 5
eng = DbmEngine(V, [H1, H2], n_particles=M, batch_size=N, max_mf_updates=50, mf_tol=tol, l2=1e-7, max_norm=6.,
                sparsity_target=[0.2, 0.1], sparsity_cost=[1e-4, 5e-5])
eng.set('W', philox.tf_random_normal((V, H1), 0.01, 1337))
eng.set('W_1', philox.tf_random_normal((H1, H2), 0.01, 1111))
eng.set('v', (philox.uniform(1, 1, 0, M * V) < 0.13).reshape(M, V))
X = (philox.uniform(1, 2, 0, 4 * N * V) < 0.13).astype(np.float32).reshape(4 * N, V)
Xd = as_device(X)
eng.seed(1)
for i in range(5):
    eng.train_step(Xd, 2e-3, 0.9, k, row=(i % 4) * N)
eng.sync()
t0 = time.perf_counter(); tot = 0
for i in range(n):
    nmf, _ = eng.train_step(Xd, 2e-3, 0.9, k, row=(i % 4) * N); tot += nmf
eng.sync()
dt = (time.perf_counter() - t0) / n
print('DBM %d rows k=%d tol=%g: %.4f ms/update, mean n_mf=%.2f, chain stats %s' % (N, k, tol, dt * 1e3, tot / n, eng.chain_stats()))
