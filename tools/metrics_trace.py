#!/usr/bin/env python
"""Developer tool: 200 metric-fetch iterations (bm_rbm_train_step_metrics_async) at the north-star shape, for
`rocprofv3 --kernel-trace --stats -- python tools/metrics_trace.py`; prints the wall time per iteration as well."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.helpers import make_pair, synth_data
from boltzmann_machines_amd.engine import as_device
V, H, B = 784, 1024, 512
eng, _ = make_pair(V, H, max_batch=B, sample_v_states=True)
eng.seed(1)
X = as_device(synth_data(B, V, 0))
for mode in ('plain', 'metrics'):
    f = (lambda: eng.train_step(X, B, 0.05, 0.9, 1)) if mode == 'plain' else (lambda: eng.train_step_metrics_async(X, B, 0.05, 0.9, 1))
    for _ in range(20): f()
    if mode == 'metrics': eng.collect_metrics()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(200): f()
    if mode == 'metrics': eng.collect_metrics()
    eng.sync()
    print('%s: %.1f us per iteration' % (mode, (time.perf_counter() - t0) / 200 * 1e6))
