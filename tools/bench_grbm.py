#!/usr/bin/env python
"""BASELINE configs[2]: Gaussian-Bernoulli RBM 3072x5000, PCD-5, batch 256 on one MI355X
(= 1-layer DBM with a Gaussian visible layer and 256 persistent particles)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boltzmann_machines_amd.engine import DbmEngine, as_device
from boltzmann_machines_amd.utils import philox
V, H, N = 3072, 5000, 256
eng = DbmEngine(V, [H], v_unit=1, sample_v_states=True, n_particles=N, batch_size=N, max_mf_updates=1, l2=0.01)
eng.set('W', philox.tf_random_normal((V, H), 0.0008, 1337))
eng.set('v', philox.normal(1, 1, 0, N * V).reshape(N, V))
X = philox.normal(1, 2, 0, 2 * N * V).reshape(2 * N, V).astype(np.float32)
Xd = as_device(X)
eng.seed(1)
for i in range(3):
    eng.train_step(Xd, 5e-4, 0.9, 5, row=(i % 2) * N)
eng.sync()
t0 = time.perf_counter(); n = 20
for i in range(n):
    eng.train_step(Xd, 5e-4, 0.9, 5, row=(i % 2) * N)
eng.sync()
dt = (time.perf_counter() - t0) / n
F = 2.0 * N * V * H
print('GRBM 3072x5000 PCD-5 batch 256: %.3f ms/update = %.0f Gibbs-steps/s, %.1f TFLOP/s of (2*5+3)*F = %.1f GFLOP; W finite: %s' % (
    dt * 1e3, 5 / dt, 13 * F / dt / 1e12, 13 * F / 1e9, bool(np.isfinite(eng.get('W')).all())))
