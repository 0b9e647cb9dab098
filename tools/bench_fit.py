#!/usr/bin/env python
"""End-to-end `BernoulliRBM.fit()` rate at the north-star shape (host loop, metric fetches, scalar logs
and checkpoint included), next to bench.py's device-loop number."""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boltzmann_machines_amd import BernoulliRBM
from boltzmann_machines_amd.utils import philox
V, H, B, N = 784, 1024, 512, 51200
X = (philox.uniform(87654321, 42, 0, N * V).reshape(N, V) < 0.1307).astype(np.float32)
for every in (10, 10 ** 9):
    d = tempfile.mkdtemp()
    rbm = BernoulliRBM(n_visible=V, n_hidden=H, batch_size=B, max_epoch=1, learning_rate=0.05, momentum=0.9, l2=1e-5,
                       sample_v_states=True, random_seed=1337, verbose=False, model_path=d + '/',
                       metrics_config=dict(train_metrics_every_iter=every))
    rbm.fit(X)                               # epoch 1: upload, tuner, first checkpoint
    rbm.set_params(max_epoch=6)
    t0 = time.perf_counter(); rbm.fit(X); dt = time.perf_counter() - t0
    steps = 5 * (N // B)
    print('fit(): train metrics every %s iters: %.1f us per update = %.0f Gibbs-steps/s (5 epochs of %d updates, '
          'checkpoint per epoch; the call includes one 161 MB upload and ends when its last checkpoint is on disk)'
          % ('10' if every == 10 else 'never', 1e6 * dt / steps, steps / dt, N // B))
    rbm.set_params(max_epoch=46)             # a longer call: the upload and the closing write amortise
    t0 = time.perf_counter(); rbm.fit(X); dt = time.perf_counter() - t0
    steps = 40 * (N // B)
    print('        the same, 40 epochs in one call: %.1f us per update = %.0f Gibbs-steps/s' % (1e6 * dt / steps, steps / dt))
    shutil.rmtree(d, ignore_errors=True)
