#!/usr/bin/env python
"""Regenerates tests/golden/*.npz.

The reference (Python 2 + TensorFlow 1.3) cannot be imported in this environment, so the only
golden values that come FROM the reference are the known answers its own tests hold
(rbm/tests/test_rbm.py:64-67, utils doctests) — they are asserted literally in tests/test_oracle.py.
The fixtures written here are regression pins of the CPU oracle (oracle/bm_oracle.c) on seeded
inputs: parameters after 3 CD-k updates for the reference's test shape (12x8, 16 samples,
dropout 0.9, both samplers on: test_rbm.py:14-22), the same shape as a MultinomialRBM, a DBM train
step, and AIS values.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from boltzmann_machines_amd.utils import RNG, philox        # noqa: E402
from oracle import oracle as orc                             # noqa: E402


def rbm_case():
    V, H = 12, 8
    X = RNG(seed=1337).rand(16, V).astype(np.float32)       # the reference test's data (test_rbm.py:17)
    t = orc.OracleRBM(V, H, sample_v_states=True, sample_h_states=True, dropout=0.9)
    t.p['W'][...] = philox.tf_random_normal((V, H), 0.01, 1337)
    t.set_seed(4242)
    for _ in range(3):
        t.train_step(X[:10], 0.01, 0.9, 1)
        t.train_step(X[10:], 0.01, 0.9, 1)
    m, flip = t.metrics(X[:10], 1)
    return dict(W=t.p['W'], vb=t.p['vb'], hb=t.p['hb'], dW=t.p['dW'], q_means=t.p['q_means'], metrics=m, flip=flip,
                transform=t.transform(X[:8], 1))


def mrbm_case():
    """MultinomialRBM (rbm.py:25-65): the reference test shape with a 10-draw multinomial hidden unit"""
    V, H = 12, 8
    X = RNG(seed=1337).rand(16, V).astype(np.float32)
    t = orc.OracleRBM(V, H, sample_v_states=True, sample_h_states=True, h_unit=2, n_samples=10)
    t.p['W'][...] = philox.tf_random_normal((V, H), 0.01, 1337)
    t.set_seed(4242)
    for _ in range(3):
        t.train_step(X[:10], 0.01, 0.9, 1)
        t.train_step(X[10:], 0.01, 0.9, 1)
    w = t.chain(X[:10], 1)
    return dict(W=t.p['W'], vb=t.p['vb'], hb=t.p['hb'], dW=t.p['dW'], q_means=t.p['q_means'],
                h0_means=t.work['h0m'].copy(), h0_counts=t.work['h0s'].copy(), transform=t.transform(X[:8], 1))


def dbm_case():
    V, nh, N, M = 20, [12, 16], 10, 10
    t = orc.OracleDBM(V, nh, n_particles=M, batch_size=N, max_mf_updates=5, mf_tol=1e-5, l2=1e-3, max_norm=1.5,
                      sparsity_target=[0.2, 0.1], sparsity_cost=[1e-2, 5e-3])
    t.p['W'][...] = (philox.normal(1, 1, 0, 240) * np.float32(0.1)).reshape(20, 12)
    t.p['W_1'][...] = (philox.normal(1, 2, 0, 192) * np.float32(0.1)).reshape(12, 16)
    t.p['v'][...] = (philox.uniform(1, 3, 0, M * V) < 0.3).reshape(M, V)
    t.set_seed(7)
    X = (philox.uniform(1, 4, 0, N * V) < 0.2).astype(np.float32).reshape(N, V)
    out = []
    for _ in range(2):
        out.append(t.train_step(X, 0.05, 0.5, 2, want_msre=True))
    ais = t.ais(20, 9, 1, 2222)
    return dict(W=t.p['W'], W_1=t.p['W_1'], hb=t.p['hb'], hb_1=t.p['hb_1'], vb=t.p['vb'], v=t.p['v'], mu_1=t.p['mu_1'],
                n_mf=np.array([o[0] for o in out]), msre=np.array([o[1] for o in out], dtype=np.float32), ais=ais,
                log_proba=t.log_proba(X))


if __name__ == '__main__':
    np.savez(os.path.join(HERE, 'rbm_12x8.npz'), **rbm_case())
    np.savez(os.path.join(HERE, 'dbm_20_12_16.npz'), **dbm_case())
    np.savez(os.path.join(HERE, 'mrbm_12x8.npz'), **mrbm_case())
    print('wrote', os.listdir(HERE))
