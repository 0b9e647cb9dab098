"""Same-box A/B of library builds at the north-star shape (784 x 1024, batch 512, CD-1).

usage: python tools/ab_rbm.py <checkout root> [label]
Each checkout (a git worktree copied to tools/ab/<sha>/ with its own built libbm355.so) is timed in its own
process through its own Python package: 5 regions of 2000 updates (HIP events of the engine + host clock), one
driver-length region of 20 updates, then the per-kernel event times of 500 profiled updates."""
import os
import sys
import time

root = os.path.abspath(sys.argv[1])
label = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(root)
sys.path.insert(0, root)
os.chdir(root)
import numpy as np                                                      # noqa: E402
from boltzmann_machines_amd.engine import RbmEngine                      # noqa: E402
from boltzmann_machines_amd._ffi import DeviceArray                      # noqa: E402

V, H, B, NB = 784, 1024, 512, 100
rng = np.random.RandomState(0)
eng = RbmEngine(V, H, max_batch=B, l2=1e-5, sample_v_states=True, sample_h_states=True)
eng.set('W', (rng.randn(V, H) * 0.01).astype(np.float32))
X = (rng.rand(B * NB, V) < 0.1307).astype(np.float32)
Xd = DeviceArray.from_numpy(X, np.float32)
eng.seed(1337)


def region(n_epochs, nb=NB):
    eng.sync()
    t0 = time.perf_counter()
    eng.timer_start()
    for _ in range(n_epochs):
        eng.train_epoch(Xd, B * nb, B, 0.05, 0.9, 1)
    ev = eng.timer_stop()
    eng.sync()
    return ev * 1e3 / (n_epochs * nb), (time.perf_counter() - t0) * 1e6 / (n_epochs * nb)


region(5)                                  # tuner + warm-up
long_ = [region(20) for _ in range(5)]
region(1)
short = [region(1, 20) for _ in range(5)]
out = '%-10s 2000-step us/update (events): %s | wall: %s || 20-step wall: %s' % (
    label, ' '.join('%.2f' % e for e, _ in long_), ' '.join('%.2f' % w for _, w in long_),
    ' '.join('%.2f' % w for _, w in short))
try:
    eng.profile(True)
    for _ in range(5):
        eng.train_epoch(Xd, B * NB, B, 0.05, 0.9, 1)
    eng.sync()
    kt = eng.kernel_times()
    out += ' || kernels us: ' + ' '.join('%s=%.2f' % (k, 1e3 * ms / max(n, 1)) for k, (ms, n) in kt.items() if n)
except Exception as e:                      # noqa: BLE001
    out += ' || kernel_times unavailable: %r' % (e,)
print(out, flush=True)
