#!/usr/bin/env python
"""End-to-end `DBM.fit()` rate at BASELINE configs[3] (784-512-1024, 512 rows + 512 particles, PCD-5, mean-field <= 50 sweeps at
tol 1e-7) through the PUBLIC classes - host loop, schedules, the progress line's metrics and the per-epoch checkpoint included -
next to bench.py's device-loop number for the same update (`--config dbm`)."""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boltzmann_machines_amd import BernoulliRBM, DBM
from boltzmann_machines_amd.utils import philox

V, H1, H2, B, N = 784, 512, 1024, 512, 10240
X = (philox.uniform(87654321, 42, 0, N * V).reshape(N, V) < 0.1307).astype(np.float32)
d = tempfile.mkdtemp()
kw = dict(batch_size=B, max_epoch=1, learning_rate=0.05, momentum=0.9, random_seed=1337, verbose=False)
rbm1 = BernoulliRBM(n_visible=V, n_hidden=H1, dbm_first=True, model_path=d + '/r1/', **kw)
rbm1.fit(X[:2 * B])
Q = rbm1.transform(X[:2 * B])
rbm2 = BernoulliRBM(n_visible=H1, n_hidden=H2, dbm_last=True, model_path=d + '/r2/', **kw)
rbm2.fit(Q)
for every in (10, 10 ** 9):
    dbm = DBM(rbms=[rbm1, rbm2], n_particles=B, batch_size=B, n_gibbs_steps=5, max_mf_updates=50, mf_tol=1e-7,
              learning_rate=2e-3, momentum=0.9, max_epoch=1, l2=1e-7, max_norm=6., sparsity_target=[0.2, 0.1],
              sparsity_cost=[1e-4, 5e-5], random_seed=1, verbose=False, train_metrics_every_iter=every, model_path=d + '/dbm%d/' % (every % 7))
    dbm.fit(X)                               # epoch 1: upload, tuner, first checkpoint
    dbm.set_params(max_epoch=4)
    t0 = time.perf_counter(); dbm.fit(X); dt = time.perf_counter() - t0
    steps = 3 * (N // B)
    print('DBM.fit(): train metrics every %s iters: %.3f ms per update (3 epochs of %d updates, checkpoint per epoch)'
          % ('10' if every == 10 else 'never', 1e3 * dt / steps, N // B))
shutil.rmtree(d, ignore_errors=True)
