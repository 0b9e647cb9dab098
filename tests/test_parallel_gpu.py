"""-m gpu: the data-parallel split of the HIP engine with world_size 2 — two processes, each with its
own bm_rbm handle (both on the one GPU of the test box), gloo all-reduce of the fused `grad` buffer
staged through the host (RCCL refuses two ranks on one device; the driver's multi-GPU bench runs the
RCCL path).  Checked: replicas stay identical, the update equals the oracle's two-shard algebra BIT
FOR BIT, and the sample bitmaps (functions of the GLOBAL row) equal a single-process full-batch run."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V, H, BL, K, WORLD = 100, 52, 24, 2, 2
KW = dict(sample_v_states=True, l2=1e-3, sparsity_cost=1e-2)


def _inputs():
    from oracle import oracle as orc
    W = (orc.normal(1, 2, 0, V * H) * np.float32(0.1)).reshape(V, H)
    Xg = (orc.uniform(1, 3, 0, WORLD * BL * V) < 0.3).astype(np.float32).reshape(WORLD * BL, V)
    return W, Xg


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from boltzmann_machines_amd import _ffi, parallel
    from boltzmann_machines_amd.engine import RbmEngine, as_device
    dist.init_process_group('gloo', rank=rank, world_size=world)
    W, Xg = _inputs()
    eng = RbmEngine(V, H, max_batch=BL, **KW)
    eng.set('W', W)
    eng.seed(99)
    grad = eng.device_view('grad')

    def allreduce_():
        eng.sync()
        host = grad.numpy()
        dist.all_reduce(torch.from_numpy(host))
        _ffi.check(_ffi.load().bm_h2d(grad.ptr, host.ctypes.data_as(C.c_void_p), host.nbytes))
    dp = parallel.DataParallelRBM(eng, rank, world, BL, allreduce_)
    Xd = as_device(Xg[rank * BL:(rank + 1) * BL])
    for step in range(3):
        dp.train_step(Xd, 0.05, 0.5, K)
    eng.sync()
    np.savez(out + '.r%d' % rank, **{n: eng.get(n) for n in ('W', 'vb', 'hb', 'dW', 'q_means')})
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_dp_world2_on_gpu(gpu_lib, tmp_path):
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    out = str(tmp_path / 'dp')
    mp.spawn(_worker, args=(WORLD, _free_port(), out), nprocs=WORLD, join=True)
    r0, r1 = np.load(out + '.r0.npz'), np.load(out + '.r1.npz')
    for n in r0.files:                                         # replicas identical
        assert np.array_equal(r0[n].view(np.uint32), r1[n].view(np.uint32)), n
    # the oracle's two-shard algebra: raw sums per shard (global row offsets), summed, applied with N = 2*BL
    W, Xg = _inputs()
    twins = []
    for r in range(WORLD):
        t = orc.OracleRBM(V, H, **KW)
        t.p['W'][...] = W
        t.set_seed(99)
        t.row0 = r * BL
        twins.append(t)
    for step in range(3):
        raws = [t.raw_grads(Xg[r * BL:(r + 1) * BL], K) for r, t in enumerate(twins)]
        total = raws[0] + raws[1]
        for t in twins:
            t.apply(total, float(WORLD * BL), 0.05, 0.5)
    for n in ('W', 'vb', 'hb', 'dW', 'q_means'):
        assert np.array_equal(r0[n].view(np.uint32), twins[0].p[n].view(np.uint32)), n
    # and the same model trained on the full batch by one engine agrees to fp32 round-off
    # (the per-rank sums are blocked differently), i.e. the rank count only changes the rounding
    ref = orc.OracleRBM(V, H, **KW)
    ref.p['W'][...] = W
    ref.set_seed(99)
    for step in range(3):
        ref.train_step(Xg, 0.05, 0.5, K)
    np.testing.assert_allclose(r0['W'], ref.p['W'], rtol=2e-5, atol=2e-7)


NATIVE_SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r)
import ctypes as C
import numpy as np
assert 'torch' not in sys.modules
from boltzmann_machines_amd import _ffi, parallel
from boltzmann_machines_amd._ffi import DeviceArray
from boltzmann_machines_amd.engine import RbmEngine, as_device
from tests.test_parallel_gpu import _inputs, V, H, BL, K, KW
W, Xg = _inputs()
comm = parallel.NativeComm(0, 1, parallel.NativeComm.unique_id())
e1 = RbmEngine(V, H, max_batch=BL, **KW)
e2 = RbmEngine(V, H, max_batch=BL, **KW)
for e in (e1, e2):
    e.set('W', W); e.seed(7)
dp = parallel.DataParallelRBM(e2, 0, 1, BL, parallel.native_allreduce_on_engine_stream(e2, comm))
Xd = as_device(Xg[:BL])
for step in range(3):
    e1.train_step(Xd, BL, 0.05, 0.5, K)
    dp.train_step(Xd, 0.05, 0.5, K)
for n in ('W', 'vb', 'hb', 'dW', 'q_means'):
    assert np.array_equal(e1.get(n).view(np.uint32), e2.get(n).view(np.uint32)), n
a = DeviceArray.from_numpy(np.arange(10, dtype=np.float32))
b = DeviceArray((10,))
_ffi.check(_ffi.load().bm_comm_allgather(comm._c, a.ptr, b.ptr, 10, C.c_void_p(e2.stream())))
e2.sync()
assert np.array_equal(b.numpy(), np.arange(10, dtype=np.float32))
comm.close(); e1.close(); e2.close()
assert 'torch' not in sys.modules
print('NATIVE_COMM_OK')
"""


def test_native_rccl_comm_world1(gpu_lib):
    """the library's own RCCL communicator (bm_comm_*), world 1 on the test box, in a process that never
    imports torch: init, the all-reduce of the fused grad buffer on the engine stream between grad_step and
    apply_step (== the fused train_step bit for bit), all-gather, destroy."""
    import subprocess
    r = subprocess.run([sys.executable, '-c', NATIVE_SCRIPT % dict(root=ROOT)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and 'NATIVE_COMM_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


NATIVE_DBM_SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
assert 'torch' not in sys.modules
from boltzmann_machines_amd import parallel
from boltzmann_machines_amd.engine import as_device
from tests import test_dbm_parity_gpu as D
V, nh, N, M = 20, [12, 16], 10, 10
kw = dict(max_mf_updates=5, mf_tol=1e-5, l2=1e-3, max_norm=1.5, sparsity_target=[0.2, 0.1], sparsity_cost=[1e-2, 5e-3])
comm = parallel.NativeComm(0, 1, parallel.NativeComm.unique_id())
e1, _ = D.make_pair(V, nh, N, M, **kw)
e2, _ = D.make_pair(V, nh, N, M, **kw)
e1.seed(42); e2.seed(42)
dp = parallel.DataParallelDBM(e2, 0, 1, parallel.native_allreduce_on_engine_stream(e2, comm), comm=comm)
for s in range(3):
    Xd = as_device(D.data(N, V, s))
    n1, _ = e1.train_step(Xd, 0.05, 0.5, 2)
    n2 = dp.train_step(Xd, 0.05, 0.5, 2)      # MF residual all-reduced (max) on the device, grads all-reduced (sum)
    assert n1 == n2, (n1, n2)
for nm in ('W', 'W_1', 'dW', 'hb', 'hb_1', 'vb', 'q_means', 'mu_means_1', 'v', 'h_1', 'mu', 'mu_1'):
    assert np.array_equal(e1.get(nm).view(np.uint32), e2.get(nm).view(np.uint32)), nm
# validation fetch and inference also run with the communicator installed
assert e1.metrics(Xd, 2) == e2.metrics(Xd, 2)
# chain-sharded AIS: shard + all-gather inside the library == the plain run
a = e1.ais(20, 37, 1, 2222)
b = e2.ais_sharded(comm, 20, 37, 1, 2222)
assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
e2.set_comm(None)
comm.close(); e1.close(); e2.close()
assert 'torch' not in sys.modules
print('NATIVE_DBM_OK')
"""


def test_native_comm_dbm_and_ais_world1(gpu_lib):
    """DBM data-parallel step and chain-sharded AIS through the library's own RCCL communicator (world 1 on the
    test box, torch-free process): device-side all-reduce(max) of the mean-field residual per sweep, one
    all-reduce(sum) of the fused gradient buffer, one all-gather of the AIS values — bit-identical to the
    fused single-GPU entry points."""
    import subprocess
    r = subprocess.run([sys.executable, '-c', NATIVE_DBM_SCRIPT % dict(root=ROOT)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and 'NATIVE_DBM_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


DELAYED_SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
assert 'torch' not in sys.modules
from boltzmann_machines_amd import parallel
from boltzmann_machines_amd.engine import RbmEngine, as_device
from oracle import oracle as orc
V, H, B, k, steps = 48, 40, 24, 1, 5
kw = dict(l2=1e-3, sample_v_states=True, sample_h_states=True)
W = (orc.normal(1, 2, 0, V * H) * np.float32(0.1)).reshape(V, H)
eng = RbmEngine(V, H, max_batch=B, **kw)
twin = orc.OracleRBM(V, H, **kw)
eng.set('W', W); twin.p['W'][...] = W
eng.seed(5); twin.set_seed(5)
comm = parallel.NativeComm(0, 1, parallel.NativeComm.unique_id())
dp = parallel.DelayedDataParallelRBM(eng, 0, 1, B, comm=comm)
pending = None
for s in range(steps):
    X = (orc.uniform(1, 50 + s, 0, B * V) < 0.3).astype(np.float32).reshape(B, V)
    dp.train_step(as_device(X), 0.05, 0.5, k)
    raw = twin.raw_grads(X, k)                 # the oracle on the same delayed schedule
    if pending is not None:
        twin.apply(pending, float(B), 0.05, 0.5)
    pending = raw
dp.flush(); twin.apply(pending, float(B), 0.05, 0.5)
for nm in ('W', 'dW', 'vb', 'hb', 'dvb', 'dhb', 'q_means'):
    assert np.array_equal(eng.get(nm).view(np.uint32), twin.p[nm].view(np.uint32)), nm
comm.close(); eng.close()
print('DELAYED_OK')
"""


def test_delayed_gradient_dp_world1(gpu_lib):
    """the NON-parity delayed-gradient mode through the library (two gradient slots, the reduction on the
    communication stream, bm_rbm_wait_grads): bit-identical to the oracle driven on the same delayed schedule"""
    import subprocess
    r = subprocess.run([sys.executable, '-c', DELAYED_SCRIPT % dict(root=ROOT)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and 'DELAYED_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
