import os, sys, faulthandler
faulthandler.dump_traceback_later(120, exit=True)
os.environ.setdefault('BM355_DEBUG', 'dbm_chain=2')
sys.path.insert(0, os.getcwd())
import numpy as np
from tests.test_dbm_parity_gpu import make_pair, data, assert_equal
from boltzmann_machines_amd.engine import as_device
V, h1, h2, N, M, k, mx = [int(x) for x in sys.argv[1:8]]
tol = float(sys.argv[8]); sv, s0, s1 = [bool(int(x)) for x in sys.argv[9:12]]
kw = dict(max_mf_updates=mx, mf_tol=tol, l2=1e-4, max_norm=3.0, sample_v_states=sv, sample_h_states=[s0, s1])
eng, twin = make_pair(V, [h1, h2], N, M, seed=55, **kw)
eng.seed(305); twin.set_seed(305)
trips = []
for s in range(3):
    X = data(N, V, 5 + s)
    print('update', s, flush=True)
    n1, _ = eng.train_step(as_device(X), 0.03, 0.6, k)
    print('  engine', n1, flush=True)
    n2, _ = twin.train_step(X, 0.03, 0.6, k)
    assert n1 == n2, (n1, n2)
    trips.append(n1)
assert_equal(eng, twin, ['W', 'W_1', 'v', 'h', 'h_1', 'mu', 'mu_1'])
print('OK', sys.argv[1:], trips, eng.chain_stats(), flush=True)
