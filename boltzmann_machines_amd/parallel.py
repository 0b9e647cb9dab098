"""Multi-GPU execution of the two paths that shard, one process per GPU.

* data-parallel CD-k: every rank runs the Gibbs chain and the raw outer products on its
  rows (`grad_step`), ONE all-reduce(sum) of the fused buffer [dW | sum(X-v) | sum(h0-hk)
  | sum(hk)] (RCCL over xGMI on the GPU box, enqueued on the engine's own HIP stream),
  then every rank applies the identical update with the GLOBAL batch size (`apply_step`),
  so the replicas stay bit-identical.  Sample bitmaps are a function of the GLOBAL row
  index (`set_row_offset`), i.e. independent of the number of ranks.
* AIS: the chains are independent; each rank runs a contiguous slice of the chains (the
  chain index is global in the RNG stream) and the per-chain log-weights are all-gathered
  once at the end (`log_Z()` returns them, reference dbm.py:922-923,939).

The collectives are injected (`allreduce_`, `allgather`), so the same code runs over
torch.distributed/RCCL on the GPU box and over gloo in the CPU tests.
"""
import os

import numpy as np


def shard(n, rank, world):
    """contiguous slice [start, stop) of n items for `rank` (remainder spread over the first ranks)"""
    q, r = divmod(n, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def dist_env():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


_default_comm = [None, False]


def default_comm():
    """process-wide communicator of a one-process-per-GPU job (NativeComm.from_env, created once); None at world 1"""
    if not _default_comm[1]:
        _default_comm[0] = NativeComm.from_env()
        _default_comm[1] = True
    return _default_comm[0]


class DataParallelRBM(object):
    """Drives one rank's engine (RbmEngine, or any object with grad_step / apply_step /
    set_row_offset) through data-parallel CD-k updates."""

    def __init__(self, engine, rank, world, local_batch, allreduce_, fused=None):
        self.engine, self.rank, self.world, self.local_batch = engine, rank, world, local_batch
        self.allreduce_ = allreduce_            # in-place sum over ranks of this rank's grad buffer
        # fused: a DirectExchange whose exchange_apply() does the all-reduce AND the update in one kernel (the momentum
        # buffer dW then lives slice-wise on its owners: call fused.gather_dw() before reading it)
        self.fused = fused if (fused is not None and engine.H % 4 == 0) else None
        engine.set_row_offset(rank * local_batch)

    def train_step(self, X_local, lr, momentum, k, **kw):
        self.engine.grad_step(X_local, self.local_batch, k, **kw)
        if self.fused is not None:          # exchange + update in one launch (DirectExchange.exchange_apply)
            self.fused.exchange_apply(self.local_batch * self.world, lr, momentum)
            return
        self.allreduce_()
        self.engine.apply_step(self.local_batch * self.world, lr, momentum)


class DelayedDataParallelRBM(object):
    """Data-parallel CD-k with the all-reduce OFF the critical path - a documented NON-parity mode: the parameter
    update applied at the end of step t is the reduced gradient of step t-1 (the reference, and DataParallelRBM, are
    synchronous).  The reduction of step t's gradient runs on the engine's communication stream under step t+1's
    Gibbs chain; the engine keeps two gradient slots.  `flush()` applies the gradient still in flight (call it at
    the end of a run).  With `start_reduce(slot)` / `finish_reduce(slot)` injected the same schedule runs over any
    collective (the gloo tests); by default they are the library's own communicator."""

    def __init__(self, engine, rank, world, local_batch, comm=None, start_reduce=None, finish_reduce=None):
        self.engine, self.rank, self.world, self.local_batch = engine, rank, world, local_batch
        self.t, self.pending = 0, None           # pending = (slot, lr, momentum) of the gradient in flight
        engine.set_row_offset(rank * local_batch)
        if start_reduce is None:
            start_reduce = lambda slot: engine.allreduce_grads_async(comm)
            finish_reduce = lambda slot: engine.wait_grads(slot)
        self.start_reduce, self.finish_reduce = start_reduce, finish_reduce

    def _apply_pending(self):
        if self.pending is not None:
            slot, lr, momentum = self.pending
            self.finish_reduce(slot)
            self.engine.set_grad_slot(slot)
            self.engine.apply_step(self.local_batch * self.world, lr, momentum)
            self.pending = None

    def train_step(self, X_local, lr, momentum, k, **kw):
        slot = self.t & 1
        self.engine.set_grad_slot(slot)
        self.engine.grad_step(X_local, self.local_batch, k, **kw)     # on the parameters BEFORE the pending update
        prev = self.pending
        self.pending = None
        self.start_reduce(slot)
        if prev is not None:
            self.pending = prev
            self._apply_pending()
        self.pending = (slot, lr, momentum)
        self.t += 1

    def flush(self):
        self._apply_pending()


def torch_allreduce_on_engine_stream(engine, device, group=None):
    """all-reduce of the engine's "grad" buffer by RCCL through torch.distributed (`group`: a nccl process group,
    default: the default group), enqueued on the engine's HIP stream (zero-copy: the buffer is wrapped through
    __cuda_array_interface__)."""
    import torch
    import torch.distributed as dist
    stream = torch.cuda.ExternalStream(engine.stream(), device=device)
    buf = torch.as_tensor(engine.device_view('grad'), device=device)

    def allreduce_():
        with torch.cuda.stream(stream):
            dist.all_reduce(buf, group=group)
    return allreduce_


def ais_sharded(run_ais, n_runs, rank, world, allgather):
    """run_ais(n_local, chain0) -> per-chain log Z estimates of chains [chain0, chain0+n_local);
    allgather(local ndarray, counts) -> concatenation over ranks.  Returns all n_runs values."""
    start, stop = shard(n_runs, rank, world)
    local = np.ascontiguousarray(run_ais(stop - start, start), dtype=np.float32)
    counts = [shard(n_runs, r, world)[1] - shard(n_runs, r, world)[0] for r in range(world)]
    return allgather(local, counts)


def torch_allgather(device=None):
    import torch
    import torch.distributed as dist

    def allgather(local, counts):
        m = max(counts)
        pad = np.zeros(m, dtype=np.float32)
        pad[:len(local)] = local
        t = torch.from_numpy(pad)
        if device is not None:
            t = t.to(device)
        outs = [torch.empty_like(t) for _ in counts]
        dist.all_gather(outs, t)
        return np.concatenate([o.cpu().numpy()[:c] for o, c in zip(outs, counts)])
    return allgather


class DataParallelDBM(object):
    """Data-parallel DBM update (BASELINE configs[3]): rank r owns rows [r*N_local, ...) of every
    minibatch (its own mean-field parameters) and particles [r*M_local, ...).  Per update:
    `grad_step` (mean-field with an all-reduce(max) of the residual per sweep, PCD, raw sums) ->
    ONE all-reduce(sum) of the fused buffer -> `apply_step` with the global N and M."""

    def __init__(self, engine, rank, world, allreduce_, allreduce_max=None, comm=None, xchg=None, fused=None):
        self.engine, self.rank, self.world = engine, rank, world
        self.allreduce_ = allreduce_
        # fused: a DirectExchange whose exchange_apply() replaces the all-reduce AND apply_step (column-sliced ownership;
        # the momentum buffers then live column-wise on their owners: fused.gather_dw() before they are read)
        self.fused = fused if (fused is not None and fused.fused_ok()) else None
        engine.set_row_offset(rank * engine.N, rank * engine.M)
        if xchg is not None:
            # the direct peer-memory exchange: one 8-byte store per peer and sweep, on the device, in stream order
            engine.set_xchg(xchg)
        elif comm is not None:
            # the library's own communicator: residual all-reduce(max) on the device, in stream order
            engine.set_comm(comm)
        elif allreduce_max is not None and world > 1:
            # collectives the library does not own (torch.distributed / gloo): host callback per sweep
            engine.set_mf_allreduce(allreduce_max)

    def train_step(self, X_local, lr, momentum, k, **kw):
        n_mf = self.engine.grad_step(X_local, k, **kw)
        if self.fused is not None:
            self.fused.exchange_apply(self.engine.N * self.world, lr, momentum, M_global=self.engine.M * self.world)
            return n_mf
        self.allreduce_()
        self.engine.apply_step(self.engine.N * self.world, self.engine.M * self.world, lr, momentum)
        return n_mf


def torch_allreduce_max():
    import torch
    import torch.distributed as dist

    def allreduce_max(x):
        t = torch.tensor([x], dtype=torch.float32)
        if dist.get_backend() == 'nccl':
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return allreduce_max


class NativeComm(object):
    """RCCL communicator owned by libbm355.so (`bm_comm_*`, include/bm355.h): the all-reduce of the fused
    `grad` buffer is enqueued on the engine's HIP stream by the library itself - no torch in the data path.
    Only the 128-byte id travels through the host (here: any initialised torch.distributed backend,
    e.g. gloo; MPI or a shared file would do as well)."""

    def __init__(self, rank, world, id_bytes):
        import ctypes as C
        from . import _ffi
        self._ffi, self.rank, self.world = _ffi, rank, world
        buf = (C.c_char * 128).from_buffer_copy(bytes(id_bytes))
        self._c = C.c_void_p()
        _ffi.check(_ffi.load().bm_comm_init(rank, world, buf, C.byref(self._c)))

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _ffi
        buf = (C.c_char * 128)()
        _ffi.check(_ffi.load().bm_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_env(cls):
        """communicator of a job launched one process per GPU (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the
        environment, e.g. by torch.distributed.run); the 128-byte id travels over torch.distributed when a process
        group exists, else over a plain TCP socket on MASTER_PORT + 1.  None when WORLD_SIZE <= 1."""
        rank, _, world = dist_env()
        if world <= 1:
            return None
        try:
            import sys
            dist = sys.modules.get('torch.distributed')
            if dist is not None and dist.is_initialized():
                return cls.from_torch_rendezvous(rank, world)
        except Exception:
            pass
        return cls(rank, world, socket_broadcast(cls.unique_id() if rank == 0 else None, rank, world))

    @classmethod
    def from_torch_rendezvous(cls, rank, world):
        """rank 0 creates the id, torch.distributed (already initialised, any backend) broadcasts it"""
        import torch.distributed as dist
        box = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        return cls(rank, world, box[0])

    def allreduce_max(self, darr, count):
        """in-place all-reduce(max) of `count` floats of a DeviceArray (default stream; synchronises)"""
        lib = self._ffi.load()
        self._ffi.check(lib.bm_comm_allreduce_max(self._c, darr.ptr, count, None))    # null stream; a later blocking
                                                                                      # copy (darr.numpy()) orders behind it

    def allreduce_grads(self, engine):
        """in-place all-reduce(sum) of the engine's fused grad buffer on the engine's stream"""
        from .engine import DbmEngine
        lib = self._ffi.load()
        f = lib.bm_dbm_allreduce_grads if isinstance(engine, DbmEngine) else lib.bm_rbm_allreduce_grads
        self._ffi.check(f(engine._h, self._c))

    def close(self):
        if getattr(self, '_c', None) is not None and self._c:
            self._ffi.load().bm_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DirectExchange(object):
    """One-shot all-reduce over peer-mapped device memory (`bm_xchg_*`, include/bm355.h): every rank maps every
    peer's `grad` buffer through hipIpc handles and ONE kernel per rank does reduce-scatter + all-gather straight
    over the xGMI links (sums in rank order 0..N-1: all replicas receive the same bits).  The host only gathers the
    256-byte blobs once, at construction: over torch.distributed when a process group exists, else over TCP
    (`socket_allgather`).  One process per rank; several ranks may share one device (tests on a 1-GPU box): pass
    `max_workgroups` (e.g. 48) then, see bm_xchg_set_max_workgroups."""

    def __init__(self, engine, rank, world, gather=None, max_workgroups=None):
        import ctypes as C
        from . import _ffi
        from .engine import DbmEngine
        self._ffi, self.engine, self.rank, self.world = _ffi, engine, rank, world
        lib = _ffi.load()
        self._dbm = isinstance(engine, DbmEngine)
        self._c = C.c_void_p()
        # Set-up is COLLECTIVE and symmetric: every rank reaches the one gather, carrying its blob or its error; if
        # any rank failed - before or after the gather - every rank raises (the second, cheap gather confirms the
        # attach), so no rank is left waiting in a collective another rank never enters.
        err, blob = None, None
        try:
            create = lib.bm_dbm_xchg_create if self._dbm else lib.bm_rbm_xchg_create
            _ffi.check(create(engine._h, rank, world, C.byref(self._c)))
            if max_workgroups is not None:
                # ranks that share ONE device (dry runs): exchange workgroups spinning on every CU can keep the peer
                # process's kernels from being placed - bound the launch well below the CU count
                _ffi.check(lib.bm_xchg_set_max_workgroups(self._c, int(max_workgroups)))
            if world > 1:
                buf = (C.c_char * 256)()
                _ffi.check(lib.bm_xchg_export(self._c, buf))
                blob = bytes(buf)
        except Exception as e:      # noqa: BLE001
            err = '%s: %s' % (type(e).__name__, e)
        self._gather = None
        if world > 1:
            gather = gather or default_gather(rank, world)
            self._gather = gather
            got = gather((err, blob))
            if err is None and not any(g[0] for g in got):
                try:
                    blobs = [g[1] for g in got]
                    assert len(blobs) == world and all(b is not None and len(b) == 256 for b in blobs)
                    allb = (C.c_char * (256 * world)).from_buffer_copy(b''.join(blobs))
                    _ffi.check(lib.bm_xchg_attach(self._c, allb))
                except Exception as e:      # noqa: BLE001
                    err = '%s: %s' % (type(e).__name__, e)
            errs = [g[0] for g in got if g[0]] + ([err] if err and not any(g[0] for g in got) else [])
            confirm = gather((err if err else None, None))
            errs = [c[0] for c in confirm if c[0]] or errs
            if errs:
                self.close()
                raise _ffi.Bm355Error('direct exchange set-up failed on %d rank(s): %s' % (len(errs), errs[0]))
        elif err:
            raise _ffi.Bm355Error(err)

    def allreduce_grads(self, engine=None):
        """in-place all-reduce(sum) of the engine's fused grad buffer, one kernel on the engine's stream"""
        lib = self._ffi.load()
        f = lib.bm_dbm_allreduce_grads_direct if self._dbm else lib.bm_rbm_allreduce_grads_direct
        self._ffi.check(f(self.engine._h, self._c))

    def exchange_apply(self, B_global, lr, momentum, M_global=None):
        """reduce-scatter + parameter update on the owned slice + all-gather of the new weights on the engine's stream:
        replaces allreduce_grads() + engine.apply_step(), same bits.  RBM (bm_rbm_exchange_apply_direct): one kernel,
        contiguous slices of W.  DBM (bm_dbm_exchange_apply_direct; M_global = the global number of particles): column
        slices of every W_i, the max-norm rescale on the owned columns between the two launches."""
        lib = self._ffi.load()
        if self._dbm and M_global is None:
            raise ValueError('exchange_apply of a DBM needs M_global, the global number of particles')
        if self._dbm:
            self._ffi.check(lib.bm_dbm_exchange_apply_direct(self.engine._h, self._c, int(B_global), int(M_global), lr, momentum))
        else:
            self._ffi.check(lib.bm_rbm_exchange_apply_direct(self.engine._h, self._c, int(B_global), lr, momentum))

    def fused_ok(self):
        """whether exchange_apply() can serve this engine (RBM: n_hidden % 4 == 0; DBM: every hidden width % 4 == 0)"""
        if not self._dbm:
            return self.engine.H % 4 == 0
        import ctypes as C
        ok = C.c_int32()
        self._ffi.check(self._ffi.load().bm_dbm_exchange_apply_ok(self.engine._h, self._c, C.byref(ok)))
        return bool(ok.value)

    def gather_dw(self):
        """complete every replica's momentum buffer dW (owners hold their slices between updates).  COLLECTIVE: every rank
        must call it, whether or not its own copy is stale (a `set('dW', ...)` must likewise be made on every rank)"""
        lib = self._ffi.load()
        f = lib.bm_dbm_exchange_gather_dw if self._dbm else lib.bm_rbm_exchange_gather_dw
        self._ffi.check(f(self.engine._h, self._c))

    def set_timeout(self, seconds):
        """bound of every in-kernel wait of later launches (a wait that expires is FATAL: the status word is sticky, the
        results are NaN-poisoned and the engine's sync() raises)"""
        self._ffi.check(self._ffi.load().bm_xchg_set_timeout(self._c, float(seconds)))

    def status(self):
        """0 when no in-kernel wait has timed out (synchronises the device)"""
        import ctypes as C
        st = C.c_int32()
        self._ffi.check(self._ffi.load().bm_xchg_status(self._c, C.byref(st)))
        return int(st.value)

    def close(self, barrier=True):
        """Frees the exchange.  COLLECTIVE when world > 1 and a gather channel is known: every rank synchronises its
        device and meets the others once more before anything is unmapped - a rank's last kernel ends when its peers have
        published DONE, not when they have finished pulling its staging slice (round-3 advisor)."""
        if getattr(self, '_c', None) is not None and self._c:
            if barrier and self.world > 1 and getattr(self, '_gather', None) is not None:
                try:
                    self.engine.sync()
                    self._gather((None, None))
                except Exception:       # noqa: BLE001 - a lost peer must not keep this rank from freeing its side
                    pass
            self._ffi.load().bm_xchg_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close(barrier=False)       # garbage collection is not a collective moment
        except Exception:
            pass


def default_gather(rank, world):
    """bytes -> list of every rank's bytes: torch.distributed when initialised, else TCP through rank 0"""
    import sys
    dist = sys.modules.get('torch.distributed')
    if dist is not None and dist.is_available() and dist.is_initialized():
        def gather(b):
            out = [None] * world
            dist.all_gather_object(out, b)
            return out
        return gather
    return lambda b: socket_allgather(b, rank, world)


def direct_allreduce_on_engine_stream(engine, xchg):
    """`allreduce_` for DataParallelRBM / DataParallelDBM over the direct peer-memory exchange"""
    def allreduce_():
        xchg.allreduce_grads()
    return allreduce_


_SOCKET_CALLS = [0]


_MAX_MSG = 1 << 20


def _wire_encode(x):
    """bytes / str / None / int and tuples or lists of them -> JSON text (bytes as base64): the rendezvous payloads are
    a 256-byte blob and an error string; nothing that arrives over the socket is ever unpickled or evaluated"""
    import base64
    import json

    def enc(v):
        if isinstance(v, (bytes, bytearray)):
            return {'b': base64.b64encode(bytes(v)).decode('ascii')}
        if isinstance(v, (list, tuple)):
            return {'l': [enc(y) for y in v]}
        if v is None or isinstance(v, (str, int)):
            return {'v': v}
        raise TypeError('socket rendezvous carries bytes, str, int, None and tuples of them, not %s' % type(v).__name__)
    return json.dumps(enc(x)).encode('ascii')


def _wire_decode(raw):
    import base64
    import json

    def dec(d):
        if not isinstance(d, dict) or len(d) != 1:
            raise ValueError('malformed rendezvous message')
        (k, v), = d.items()
        if k == 'b' and isinstance(v, str):
            return base64.b64decode(v.encode('ascii'), validate=True)
        if k == 'l' and isinstance(v, list):
            return tuple(dec(y) for y in v)
        if k == 'v' and (v is None or isinstance(v, (str, int))):
            return v
        raise ValueError('malformed rendezvous message')
    return dec(json.loads(raw.decode('ascii')))


def _recv_msg(c):
    """one length-prefixed message (4-byte little-endian length, at most _MAX_MSG bytes) from a socket with a timeout"""
    buf = b''
    while len(buf) < 4:
        chunk = c.recv(4 - len(buf))
        if not chunk:
            raise ConnectionError('rendezvous peer closed the connection')
        buf += chunk
    n = int.from_bytes(buf, 'little')
    if n > _MAX_MSG:
        raise ValueError('rendezvous message of %d bytes refused' % n)
    out = b''
    while len(out) < n:
        chunk = c.recv(min(65536, n - len(out)))
        if not chunk:
            raise ConnectionError('rendezvous peer closed the connection')
        out += chunk
    return out


def socket_allgather(payload, rank, world, addr=None, port=None, timeout=120.0):
    """every rank sends `payload` (bytes / str / None / tuples of them) to rank 0 over TCP and receives the tuple of
    all payloads in rank order.  Messages are length-prefixed JSON (no pickle: a peer that can reach the port can
    make the set-up FAIL, not run code); rank 0 checks the sender's rank, refuses duplicates and oversized messages,
    and every accepted connection carries the call's timeout."""
    import socket
    import time
    addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
    if port is None:
        # every call of a process takes the next port (all ranks call in the same order): a client of call n+1 must
        # not land in the backlog of call n's listening socket, which rank 0 is about to close
        port = int(os.environ.get('MASTER_PORT', '29533')) + 2 + _SOCKET_CALLS[0]
        _SOCKET_CALLS[0] += 1
    port = int(port)
    deadline = time.time() + timeout

    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world)
        parts, conns = {0: payload}, []
        try:
            while len(parts) < world:
                srv.settimeout(max(0.01, deadline - time.time()))
                c, _a = srv.accept()
                c.settimeout(max(0.01, deadline - time.time()))
                try:
                    msg = _wire_decode(_recv_msg(c))
                    if not (isinstance(msg, tuple) and len(msg) == 2 and isinstance(msg[0], int)
                            and 0 < msg[0] < world and msg[0] not in parts):
                        raise ValueError('unexpected sender')
                except (ValueError, ConnectionError, OSError):
                    c.close()               # not one of this job's ranks: drop it and keep waiting for them
                    continue
                parts[msg[0]] = msg[1]
                conns.append(c)
            out = tuple(parts[r] for r in range(world))
            wire = _wire_encode(out)
            for c in conns:
                c.sendall(len(wire).to_bytes(4, 'little') + wire)
        finally:
            for c in conns:
                c.close()
            srv.close()
        return list(out)
    while True:
        try:
            c = socket.create_connection((addr, port), timeout=5.0)
            break
        except OSError:
            if time.time() > deadline:
                raise
            time.sleep(0.05)
    try:
        c.settimeout(max(0.01, deadline - time.time()))
        wire = _wire_encode((int(rank), payload))
        c.sendall(len(wire).to_bytes(4, 'little') + wire)
        out = _wire_decode(_recv_msg(c))
    finally:
        c.close()
    if not (isinstance(out, tuple) and len(out) == world):
        raise ValueError('malformed rendezvous reply')
    return list(out)


def socket_broadcast(payload, rank, world, addr=None, port=None, timeout=120.0):
    """rank 0 sends `payload` (bytes) to every other rank over TCP (torch-free rendezvous for the RCCL id)"""
    import socket
    import time
    addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(port or int(os.environ.get('MASTER_PORT', '29533')) + 1)
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world)
        srv.settimeout(timeout)
        for _ in range(world - 1):
            c, _a = srv.accept()
            c.sendall(len(payload).to_bytes(4, 'little') + payload)
            c.close()
        srv.close()
        return payload
    t0 = time.time()
    while True:
        try:
            c = socket.create_connection((addr, port), timeout=5.0)
            break
        except OSError:
            if time.time() - t0 > timeout:
                raise
            time.sleep(0.05)
    buf = b''
    while len(buf) < 4 or len(buf) < 4 + int.from_bytes(buf[:4], 'little'):
        chunk = c.recv(4096)
        if not chunk:
            break
        buf += chunk
    c.close()
    return buf[4:4 + int.from_bytes(buf[:4], 'little')]


def native_allreduce_on_engine_stream(engine, comm):
    """`allreduce_` for DataParallelRBM / DataParallelDBM over the library's own RCCL communicator"""
    def allreduce_():
        comm.allreduce_grads(engine)
    return allreduce_
