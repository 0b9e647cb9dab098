#!/usr/bin/env python
"""float64 CD-1 update rate at the north-star shape (bm_rbm64_*, FP64 MFMA) next to the float32 engine."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from boltzmann_machines_amd._ffi import DeviceArray
from boltzmann_machines_amd.engine import RbmEngine, RbmEngine64, as_device
from boltzmann_machines_amd.utils import philox
V, H, B = 784, 1024, 512
X = (philox.uniform(87654321, 42, 0, B * V).reshape(B, V) < 0.1307)
W = philox.tf_random_normal((V, H), 0.01, 1337)
for name, E, dt in (('float64', RbmEngine64, np.float64), ('float32', RbmEngine, np.float32)):
    eng = E(V, H, max_batch=B, l2=1e-5, sample_v_states=True)
    eng.set('W', W.astype(dt)); eng.seed(1)
    Xd = DeviceArray.from_numpy(X.astype(dt), dt)
    for _ in range(50):
        eng.train_step(Xd, B, 0.05, 0.9, 1)
    eng.sync()
    t0 = time.perf_counter(); n = 300
    for _ in range(n):
        eng.train_step(Xd, B, 0.05, 0.9, 1)
    eng.sync()
    us = 1e6 * (time.perf_counter() - t0) / n
    print('%s CD-1 update 784x1024 batch 512: %.1f us = %.1f TFLOP/s' % (name, us, 4.110e9 / us / 1e6))
