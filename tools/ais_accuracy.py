#!/usr/bin/env python
"""How far the engine's AIS values are from the float64-accumulating CPU oracle on the same chains (64 chains x 1000 betas of the
784-512-1024 DBM, the setting of tests/test_full_size_gpu.py::test_config4_ais_1000_betas_vs_oracle) and from the exactly
enumerable log Z of a small DBM: the numbers behind the choice of softplus in the log-weight epilogue."""
import os
import sys
import numpy as np
os.environ.setdefault('OMP_NUM_THREADS', '16')          # the oracle is OpenMP: 256 spinning threads take minutes (tests/conftest.py)
os.environ.setdefault('OMP_WAIT_POLICY', 'passive')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_dbm_parity_gpu as D  # noqa: E402

V, nh, N = 784, [512, 1024], 64
eng, twin = D.make_pair(V, nh, N, N)
g = eng.ais(n_betas=1000, n_runs=64, k=1, seed=2222, chain0=12345).astype(np.float64)
c = twin.ais(n_betas=1000, n_runs=64, k=1, seed=2222, chain0=12345).astype(np.float64)
print('784-512-1024, 64 chains x 1000 betas: max |engine - oracle| = %.3e absolute, %.3e relative (values ~ %.1f)'
      % (np.max(np.abs(g - c)), np.max(np.abs(g - c) / np.abs(c)), np.mean(np.abs(c))))
eng.close()
