#!/usr/bin/env python
"""Timing of the DBM rows of BASELINE.json (configs[3] and [4]) on one MI355X — parity-test
configurations, not the bench.py line; numbers are quoted in DESIGN.md."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boltzmann_machines_amd.engine import DbmEngine, as_device
from boltzmann_machines_amd.utils import philox

V, H1, H2 = 784, 512, 1024
N = M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
eng = DbmEngine(V, [H1, H2], n_particles=M, batch_size=N, max_mf_updates=50, mf_tol=1e-7, l2=1e-7, max_norm=6.,
                sparsity_target=[0.2, 0.1], sparsity_cost=[1e-4, 5e-5])
eng.set('W', philox.tf_random_normal((V, H1), 0.01, 1337))
eng.set('W_1', philox.tf_random_normal((H1, H2), 0.01, 1111))
eng.set('v', (philox.uniform(1, 1, 0, M * V) < 0.13).reshape(M, V))
X = (philox.uniform(1, 2, 0, 4 * N * V) < 0.13).astype(np.float32).reshape(4 * N, V)
Xd = as_device(X)
eng.seed(1)
for i in range(3):
    nmf, _ = eng.train_step(Xd, 2e-3, 0.9, 5, row=(i % 4) * N)
eng.sync()
t0 = time.perf_counter(); n = 20; tot = 0
for i in range(n):
    nmf, _ = eng.train_step(Xd, 2e-3, 0.9, 5, row=(i % 4) * N); tot += nmf
eng.sync()
dt = (time.perf_counter() - t0) / n
T = tot / n
flops = 2*N*V*H1 + T*(4*N*H1*H2) + 5*M*(4*V*H1 + 4*H1*H2) + 2*(N+M)*(V*H1 + H1*H2)
print('DBM 784-512-1024 N=M=%d PCD-5: %.3f ms/update, mean n_mf=%.1f, %.1f TFLOP/s algorithmic' % (N, dt*1e3, T, flops/dt/1e12))
R, nb = 20000, 50
eng.ais(5, R, 1, 2222)
t0 = time.perf_counter(); vals = eng.ais(nb, R, 1, 2222); dt = time.perf_counter() - t0
print('AIS %d chains: %.3f ms per beta-step (k=1), %.1f TFLOP/s (4*M*H1*(V+H2) per step); 1000 betas ~ %.2f s; logZ mean %.3f' % (
    R, dt/nb*1e3, 4.0*R*H1*(V+H2)/(dt/nb)/1e12, dt/nb*1000, float(np.mean(vals))))
