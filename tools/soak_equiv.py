#!/usr/bin/env python
"""Developer tool: long runs whose final state must not depend on HOW the passes were launched.  Prints one CRC line per
case; run it under different switches (BM355_CHAIN=0|2, BM355_UP_XM=0|1, BM355_DBM_XM=0|1) and compare the outputs:
    for c in 0 2; do for x in 0 1; do BM355_CHAIN=$c BM355_UP_XM=$x BM355_DBM_XM=$x python tools/soak_equiv.py; done; done | sort | uniq -c"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.helpers import make_pair, synth_data
from boltzmann_machines_amd.engine import as_device, DbmEngine
from boltzmann_machines_amd._ffi import DeviceArray
from oracle import oracle as orc

crc = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes())
for V, H, B, k, steps in ((784, 1024, 512, 1, 400), (784, 1024, 512, 3, 120), (320, 448, 200, 2, 200), (1024, 768, 1100, 1, 60)):
    eng, _ = make_pair(V, H, max_batch=B, sample_v_states=True, l2=1e-5)
    eng.seed(17)
    X = as_device(np.concatenate([synth_data(B, V, s) for s in range(4)]))
    for e in range(steps // 4):
        eng.train_epoch(X, 4 * B, B, 0.05, 0.9, k)
    Hd = DeviceArray.from_numpy(synth_data(B, H, 3))
    Vd = DeviceArray((B, V))
    for r in range(20):
        eng.gibbs(Hd, Vd, B, 7)
    eng.sync()
    print('rbm %dx%d B=%d k=%d: W %08x hb %08x gibbs-h %08x gibbs-v %08x' % (V, H, B, k, crc(eng.get('W')), crc(eng.get('hb')),
                                                                              crc(Hd.numpy()), crc(Vd.numpy())))
    eng.close()
V, nh, N = 784, [512, 1024], 512
kw = dict(n_particles=N, batch_size=N, max_mf_updates=30, mf_tol=1e-7, l2=1e-4, max_norm=6.0)
eng = DbmEngine(V, nh, **kw)
W0 = (orc.normal(1, 1, 0, V * nh[0]) * np.float32(0.05)).reshape(V, nh[0])
W1 = (orc.normal(1, 2, 0, nh[0] * nh[1]) * np.float32(0.05)).reshape(nh[0], nh[1])
P0 = (orc.uniform(1, 3, 0, N * V) < 0.3).astype(np.float32).reshape(N, V)
for name, val in (('W', W0), ('W_1', W1), ('v', P0)):
    eng.set(name, val)
eng.seed(7)
X = as_device((orc.uniform(1, 4, 0, N * V) < 0.2).astype(np.float32).reshape(N, V))
for s in range(25):
    eng.train_step(X, 0.01, 0.5, 5)
eng.sync()
print('dbm 784-512-1024: ' + ' '.join('%s %08x' % (n, crc(eng.get(n))) for n in ('W', 'W_1', 'hb', 'hb_1', 'mu', 'mu_1', 'v')))
