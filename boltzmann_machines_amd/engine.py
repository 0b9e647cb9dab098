"""Thin object wrappers over the C-ABI handles (include/bm355.h).

`RbmEngine` plays the role the TF session plays in the reference: it owns the
variables of one model in HBM and executes the fetch sites of
boltzmann_machines/rbm/base_rbm.py (`session.run(train_op)` :566, metric
fetches :554-564,:578, `transform_op.eval` :697, free energy :605,:612).
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import DeviceArray, check

RBM_VARS = ('W', 'vb', 'hb', 'dW', 'dvb', 'dhb', 'q_means', 'sigma')


def as_device(X):
    """host ndarray -> DeviceArray (float32, C order); DeviceArray passes through."""
    if isinstance(X, DeviceArray):
        return X
    return DeviceArray.from_numpy(np.asarray(X), np.float32)


class RbmEngine(object):
    dtype = np.float32

    def __init__(self, n_visible, n_hidden, v_unit=_ffi.UNIT_BERNOULLI, sample_v_states=False,
                 sample_h_states=True, dbm_first=False, dbm_last=False, max_batch=10, l2=1e-4,
                 sparsity_target=0.1, sparsity_cost=0., sparsity_damping=0.9, dropout=None,
                 h_unit=_ffi.UNIT_BERNOULLI, n_samples=0):
        self.lib = _ffi.load()
        self.V, self.H, self.max_batch = int(n_visible), int(n_hidden), int(max_batch)
        cfg = _ffi.RbmConfig(self.V, self.H, int(v_unit), int(bool(sample_v_states)),
                             int(bool(sample_h_states)), int(bool(dbm_first)), int(bool(dbm_last)),
                             self.max_batch, float(l2), float(sparsity_target), float(sparsity_cost),
                             float(sparsity_damping), -1.0 if dropout is None else float(dropout),
                             int(h_unit), int(n_samples))
        self._h = C.c_void_p()
        check(self.lib.bm_rbm_create(C.byref(cfg), C.byref(self._h)))

    # -- lifetime
    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self.lib.bm_rbm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- variables
    def _size(self, name):
        return {'W': self.V * self.H, 'dW': self.V * self.H, 'vb': self.V, 'dvb': self.V, 'sigma': self.V,
                'hb': self.H, 'dhb': self.H, 'q_means': self.H,
                'grad': self.V * self.H + self.V + 2 * self.H}[name]

    def _shape(self, name):
        return (self.V, self.H) if name in ('W', 'dW') else (self._size(name),)

    def set(self, name, value):
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=np.float32), self._shape(name)))
        check(self.lib.bm_rbm_set_param(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))

    def get(self, name):
        a = np.empty(self._shape(name), dtype=np.float32)
        check(self.lib.bm_rbm_get_param(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))
        return a

    def stage(self, slot):
        """snapshot of every variable into device-side slot 0 | 1, in stream order, without a host wait (bm_rbm_stage)"""
        check(self.lib.bm_rbm_stage(self._h, int(slot)))

    def get_staged(self, slot, name):
        """one variable of a staged snapshot; waits for the staged copies only, callable from another thread"""
        a = np.empty(self._shape(name), dtype=np.float32)
        check(self.lib.bm_rbm_get_staged(self._h, int(slot), name.encode(), a.ctypes.data_as(C.c_void_p), a.size))
        return a

    def set_fast_binary(self, on, everywhere=False):
        """opt-in exact-product bf16 x 3 mode of the sampling sweep (bm_rbm_set_fast_binary): where it pays (>= 8M weights);
        everywhere=True: wherever legal (tests, measurements)"""
        check(self.lib.bm_rbm_set_fast_binary(self._h, (2 if everywhere else 1) if on else 0))

    def set_from_device(self, name, darr):
        """variable <- dense DeviceArray, asynchronously on the engine's stream (bm_rbm_set_param_dev)"""
        check(self.lib.bm_rbm_set_param_dev(self._h, name.encode(), darr.ptr, int(np.prod(darr.shape))))

    def device_view(self, name):
        p, n = C.c_void_p(), C.c_size_t()
        check(self.lib.bm_rbm_dev_ptr(self._h, name.encode(), C.byref(p), C.byref(n)))
        return DeviceArray((n.value,), np.float32, ptr=p.value, owner=self)

    # -- control
    def seed(self, seed):
        check(self.lib.bm_rbm_seed(self._h, int(seed)))

    def set_row_offset(self, row0):
        check(self.lib.bm_rbm_set_row_offset(self._h, int(row0)))

    def sync(self):
        check(self.lib.bm_rbm_sync(self._h))

    # -- fetch sites
    def train_step(self, Xd, B, lr, momentum, k, row=0):
        check(self.lib.bm_rbm_train_step(self._h, Xd.offset_ptr(row * self.V), B, lr, momentum, k))

    def train_step_metrics(self, Xd, B, lr, momentum, k, row=0):
        out = (C.c_float * 4)()
        check(self.lib.bm_rbm_train_step_metrics(self._h, Xd.offset_ptr(row * self.V), B, lr, momentum, k, out))
        return np.array(out[:], dtype=np.float32)

    MAX_PENDING_METRICS = 4096

    def train_step_metrics_async(self, Xd, B, lr, momentum, k, row=0):
        """train_step_metrics without the host wait: the values arrive with the next collect_metrics()"""
        check(self.lib.bm_rbm_train_step_metrics_async(self._h, Xd.offset_ptr(row * self.V), B, lr, momentum, k))

    def collect_metrics(self):
        """[n, 4] float32: the pending asynchronous fetches, oldest first (ONE stream synchronisation)"""
        out = np.empty((self.MAX_PENDING_METRICS, 4), dtype=np.float32)
        n = C.c_int32(0)
        check(self.lib.bm_rbm_collect_metrics(self._h, out.ctypes.data_as(C.POINTER(C.c_float)),
                                              self.MAX_PENDING_METRICS, C.byref(n)))
        return out[:n.value].copy()

    def train_epoch(self, Xd, N, batch, lr, momentum, k, row=0):
        """N rows starting at `row`, consecutive batches of `batch` rows, driven from C (no Python per batch)"""
        check(self.lib.bm_rbm_train_epoch(self._h, Xd.offset_ptr(row * self.V), N, batch, lr, momentum, k))

    def grad_step(self, Xd, B, k, row=0):
        check(self.lib.bm_rbm_grad_step(self._h, Xd.offset_ptr(row * self.V), B, k))

    def apply_step(self, B_global, lr, momentum):
        check(self.lib.bm_rbm_apply_step(self._h, B_global, lr, momentum))

    # delayed-gradient data parallelism (bm355.h: bm_rbm_set_grad_slot / _allreduce_grads_async / _wait_grads)
    def set_grad_slot(self, slot):
        check(self.lib.bm_rbm_set_grad_slot(self._h, slot))

    def allreduce_grads_async(self, comm):
        check(self.lib.bm_rbm_allreduce_grads_async(self._h, comm._c))

    def wait_grads(self, slot):
        check(self.lib.bm_rbm_wait_grads(self._h, slot))

    def transform(self, Xd, B, k, Hd, row=0, out_row=0):
        check(self.lib.bm_rbm_transform(self._h, Xd.offset_ptr(row * self.V), B, k, Hd.offset_ptr(out_row * self.H)))

    def metrics(self, Xd, B, k, row=0):
        out = (C.c_float * 4)()
        check(self.lib.bm_rbm_metrics(self._h, Xd.offset_ptr(row * self.V), B, k, out))
        return np.array(out[:], dtype=np.float32)

    def free_energy(self, Xd, B, row=0):
        out = C.c_float()
        check(self.lib.bm_rbm_free_energy(self._h, Xd.offset_ptr(row * self.V), B, C.byref(out)))
        return float(out.value)

    def gibbs(self, Hd, Vd, B, n_steps):
        check(self.lib.bm_rbm_gibbs(self._h, Hd.ptr, Vd.ptr, B, n_steps))

    def stream(self):
        p = C.c_void_p()
        check(self.lib.bm_rbm_stream(self._h, C.byref(p)))
        return p.value

    def chain_stats(self):
        """(chained launches issued, tiles they must compute, BM355_DEBUG=chain=.. mode) - csrc/bm_chain.h"""
        out = (C.c_int64 * 3)()
        check(self.lib.bm_rbm_chain_stats(self._h, out))
        return int(out[0]), int(out[1]), int(out[2])

    def profile(self, enable):
        check(self.lib.bm_rbm_profile(self._h, int(bool(enable))))

    def kernel_times(self):
        ms, n = (C.c_float * 6)(), (C.c_int32 * 6)()
        check(self.lib.bm_rbm_kernel_times(self._h, ms, n))
        names = ('act_up', 'act_down', 'grad', 'colsum', 'bias', 'other')
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(names)}

    def timer_start(self):
        check(self.lib.bm_rbm_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float()
        check(self.lib.bm_rbm_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)

    def timer_mark(self):
        check(self.lib.bm_rbm_timer_mark(self._h))

    def timer_elapsed(self):
        ms = C.c_float()
        check(self.lib.bm_rbm_timer_elapsed(self._h, C.byref(ms)))
        return float(ms.value)


class RbmEngine64(object):
    """float64 RBM handle (bm_rbm64_*, include/bm355.h): same method names as RbmEngine, float64
    device arrays and scalars.  Compatibility path for dtype='float64' models (base/mixin.py:15)."""

    dtype = np.float64

    def __init__(self, n_visible, n_hidden, v_unit=_ffi.UNIT_BERNOULLI, sample_v_states=False,
                 sample_h_states=True, dbm_first=False, dbm_last=False, max_batch=10, l2=1e-4,
                 sparsity_target=0.1, sparsity_cost=0., sparsity_damping=0.9, dropout=None,
                 h_unit=_ffi.UNIT_BERNOULLI, n_samples=0):
        self.lib = _ffi.load()
        self.V, self.H, self.max_batch = int(n_visible), int(n_hidden), int(max_batch)
        drop = -1.0 if dropout is None else float(dropout)
        cfg = _ffi.RbmConfig(self.V, self.H, int(v_unit), int(bool(sample_v_states)),
                             int(bool(sample_h_states)), int(bool(dbm_first)), int(bool(dbm_last)),
                             self.max_batch, float(l2), float(sparsity_target), float(sparsity_cost),
                             float(sparsity_damping), drop, int(h_unit), int(n_samples))
        hyper = (C.c_double * 5)(float(l2), float(sparsity_target), float(sparsity_cost), float(sparsity_damping), drop)
        self._h = C.c_void_p()
        check(self.lib.bm_rbm64_create(C.byref(cfg), hyper, C.byref(self._h)))

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self.lib.bm_rbm64_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _shape(self, name):
        if name in ('W', 'dW'):
            return (self.V, self.H)
        return (self.V,) if name in ('vb', 'dvb', 'sigma') else (self.H,)

    def set(self, name, value):
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=np.float64), self._shape(name)))
        check(self.lib.bm_rbm64_set_param(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))

    def get(self, name):
        a = np.empty(self._shape(name), dtype=np.float64)
        check(self.lib.bm_rbm64_get_param(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))
        return a

    def seed(self, seed):
        check(self.lib.bm_rbm64_seed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF))

    def set_row_offset(self, row0):
        check(self.lib.bm_rbm64_set_row_offset(self._h, int(row0)))

    def sync(self):
        check(self.lib.bm_rbm64_sync(self._h))

    def train_step(self, Xd, B, lr, momentum, k, row=0):
        check(self.lib.bm_rbm64_train_step(self._h, Xd.offset_ptr(row * self.V), B, lr, momentum, k))

    def train_step_metrics(self, Xd, B, lr, momentum, k, row=0):
        out = (C.c_double * 4)()
        check(self.lib.bm_rbm64_train_step_metrics(self._h, Xd.offset_ptr(row * self.V), B, lr, momentum, k, out))
        return np.array(out[:], dtype=np.float64)

    def transform(self, Xd, B, k, Hd, row=0, out_row=0):
        check(self.lib.bm_rbm64_transform(self._h, Xd.offset_ptr(row * self.V), B, k, Hd.offset_ptr(out_row * self.H)))

    def metrics(self, Xd, B, k, row=0):
        out = (C.c_double * 4)()
        check(self.lib.bm_rbm64_metrics(self._h, Xd.offset_ptr(row * self.V), B, k, out))
        return np.array(out[:], dtype=np.float64)

    def free_energy(self, Xd, B, row=0):
        out = C.c_double()
        check(self.lib.bm_rbm64_free_energy(self._h, Xd.offset_ptr(row * self.V), B, C.byref(out)))
        return float(out.value)


class DbmEngine(object):
    """One bm_dbm handle: the variables, variational parameters and fantasy particles
    of a DBM in HBM + the fetch sites of boltzmann_machines/dbm.py (`session.run` at
    :798,:805,:813,:869,:882,:893,:930,:954)."""

    def __init__(self, n_visible, n_hiddens, v_unit=_ffi.UNIT_BERNOULLI, sample_v_states=True,
                 sample_h_states=None, n_particles=100, batch_size=100, max_mf_updates=10, mf_tol=1e-7,
                 l2=0., max_norm=np.inf, sparsity_target=0.1, sparsity_cost=0., sparsity_damping=0.9,
                 h_units=None, n_samples=None):
        self.lib = _ffi.load()
        self.V = int(n_visible)
        self.n_hiddens = [int(x) for x in n_hiddens]
        self.L = len(self.n_hiddens)
        self.N, self.M = int(batch_size), int(n_particles)
        cfg = _ffi.DbmConfig()
        cfg.n_layers, cfg.n_visible, cfg.v_unit = self.L, self.V, int(v_unit)
        cfg.sample_v_states = int(bool(sample_v_states))
        sh = sample_h_states or [True] * self.L
        st = sparsity_target if hasattr(sparsity_target, '__iter__') else [sparsity_target] * self.L
        sc = sparsity_cost if hasattr(sparsity_cost, '__iter__') else [sparsity_cost] * self.L
        for i in range(self.L):
            cfg.n_hiddens[i] = self.n_hiddens[i]
            cfg.sample_h_states[i] = int(bool(sh[i]))
            cfg.sparsity_target[i] = float(st[i])
            cfg.sparsity_cost[i] = float(sc[i])
            cfg.h_unit[i] = int((h_units or [_ffi.UNIT_BERNOULLI] * self.L)[i])      # layers.py:39-70
            cfg.n_samples[i] = int((n_samples or [0] * self.L)[i])
        cfg.n_particles, cfg.batch_size, cfg.max_mf_updates = self.M, self.N, int(max_mf_updates)
        cfg.mf_tol, cfg.l2, cfg.max_norm = float(mf_tol), float(l2), float(max_norm)
        cfg.sparsity_damping = float(sparsity_damping)
        self._h = C.c_void_p()
        self._create(cfg, (float(mf_tol), float(l2), float(max_norm), float(sparsity_damping),
                           [float(x) for x in st], [float(x) for x in sc]))

    dtype = np.float32

    def _create(self, cfg, hyper):
        check(self.lib.bm_dbm_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            (self.lib.bm_dbm64_destroy if self.dtype == np.float64 else self.lib.bm_dbm_destroy)(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- variables ("W", "W_1", "hb", "hb_1", "mu_1", "h", "h_1", "v", ...)
    def shape(self, name):
        base, idx = name, 0
        if '_' in name and name.rsplit('_', 1)[1].isdigit():
            base, idx = name.rsplit('_', 1)[0], int(name.rsplit('_', 1)[1])
        n = [self.V] + self.n_hiddens
        if base in ('W', 'dW'):
            return (n[idx], n[idx + 1])
        if base in ('mu', 'mu_new'):
            return (self.N, n[idx + 1])
        if base in ('h', 'h_new'):
            return (self.M, n[idx + 1])
        if base in ('v', 'v_new'):
            return (self.M, self.V)
        if base in ('hb', 'dhb', 'q_means', 'mu_means', 'W_norm'):
            return (n[idx + 1],)
        if base in ('vb', 'dvb', 'sigma'):
            return (self.V,)
        raise KeyError(name)

    def set(self, name, value):
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=np.float32), self.shape(name)))
        check(self.lib.bm_dbm_set_param(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))

    def get(self, name):
        a = np.empty(self.shape(name), dtype=np.float32)
        check(self.lib.bm_dbm_get_param(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))
        return a

    def seed(self, seed):
        check(self.lib.bm_dbm_seed(self._h, int(seed)))

    def set_row_offset(self, row0, particle0):
        check(self.lib.bm_dbm_set_row_offset(self._h, int(row0), int(particle0)))

    def sync(self):
        check(self.lib.bm_dbm_sync(self._h))

    def train_step(self, Xd, lr, momentum, k, row=0, want_msre=False):
        nmf, msre = C.c_int32(), C.c_float()
        check(self.lib.bm_dbm_train_step(self._h, Xd.offset_ptr(row * self.V), lr, momentum, k, C.byref(nmf),
                                         C.byref(msre) if want_msre else None))
        return int(nmf.value), (float(msre.value) if want_msre else None)

    def metrics(self, Xd, k, row=0):
        """validation fetch (dbm.py:813): mean-field + k PCD sweeps + reconstruction msre, no update"""
        nmf, msre = C.c_int32(), C.c_float()
        check(self.lib.bm_dbm_metrics(self._h, Xd.offset_ptr(row * self.V), k, C.byref(nmf), C.byref(msre)))
        return int(nmf.value), float(msre.value)

    def grad_step(self, Xd, k, row=0):
        nmf = C.c_int32()
        check(self.lib.bm_dbm_grad_step(self._h, Xd.offset_ptr(row * self.V), k, C.byref(nmf)))
        return int(nmf.value)

    def apply_step(self, N_global, M_global, lr, momentum):
        check(self.lib.bm_dbm_apply_step(self._h, N_global, M_global, lr, momentum))

    def set_comm(self, comm):
        """install (or, with None, remove) the library-owned RCCL communicator: the mean-field residual is then
        all-reduced (max) on the device per sweep (bm_dbm_set_comm)"""
        self._comm = comm
        check(self.lib.bm_dbm_set_comm(self._h, comm._c if comm is not None else None))

    def set_fast_binary(self, on, everywhere=False):
        """opt-in exact-product bf16 x 3 mode (bm_dbm_set_fast_binary): AIS, and the particle sweeps where they gain (>= 8M
        weights in the bottom layer); everywhere=True: wherever legal (tests, measurements)"""
        check(self.lib.bm_dbm_set_fast_binary(self._h, (2 if everywhere else 1) if on else 0))

    def set_ais_literal(self, on):
        """AIS log-weights accumulated in float32 in the reference graph's order (bm_dbm_set_ais_literal)"""
        check(self.lib.bm_dbm_set_ais_literal(self._h, int(bool(on))))

    def set_sigmoid_literal(self, on):
        """every Bernoulli activation as the reference's float32 `1 / (1 + exp(-x))` (bm_dbm_set_sigmoid_literal)"""
        check(self.lib.bm_dbm_set_sigmoid_literal(self._h, int(bool(on))))

    def set_xchg(self, xchg):
        """like set_comm, with the per-sweep residual max over the direct peer-memory exchange (bm_dbm_set_xchg)"""
        self._xchg = xchg
        check(self.lib.bm_dbm_set_xchg(self._h, xchg._c if xchg is not None else None))

    def ais_sharded_direct(self, xchg, n_betas, n_runs_total, k, seed):
        """ais_sharded over the direct peer-memory exchange (parallel.DirectExchange of THIS engine) instead of the RCCL
        communicator: bm_dbm_ais_sharded_direct; returns all n_runs_total values, the owners' bits"""
        out = np.empty(n_runs_total, dtype=np.float32)
        check(self.lib.bm_dbm_ais_sharded_direct(self._h, xchg._c, n_betas, n_runs_total, k, int(seed),
                                                 out.ctypes.data_as(C.c_void_p)))
        return out

    def ais_sharded(self, comm, n_betas, n_runs_total, k, seed):
        """this rank's slice of the chains + ONE all-gather (bm_dbm_ais_sharded); returns all n_runs_total values"""
        out = np.empty(n_runs_total, dtype=np.float32)
        check(self.lib.bm_dbm_ais_sharded(self._h, comm._c, n_betas, n_runs_total, k, int(seed),
                                          out.ctypes.data_as(C.c_void_p)))
        return out

    def set_mf_allreduce(self, fn):
        """fn(local_max: float) -> global max over ranks (mean-field loop condition)"""
        self._mf_cb = _ffi.MF_REDUCE_FN(lambda x, ctx: float(fn(x))) if fn is not None else None
        check(self.lib.bm_dbm_set_mf_allreduce(self._h, C.cast(self._mf_cb, C.c_void_p) if self._mf_cb else None, None))

    def device_view(self, name):
        p, n = C.c_void_p(), C.c_size_t()
        check(self.lib.bm_dbm_dev_ptr(self._h, name.encode(), C.byref(p), C.byref(n)))
        return DeviceArray((n.value,), np.float32, ptr=p.value, owner=self)

    def stream(self):
        p = C.c_void_p()
        check(self.lib.bm_dbm_stream(self._h, C.byref(p)))
        return p.value

    def mean_field(self, Xd, row=0, out=None, out_row=0):
        nmf = C.c_int32()
        p = out.offset_ptr(out_row * self.n_hiddens[-1]) if out is not None else None
        check(self.lib.bm_dbm_mean_field(self._h, Xd.offset_ptr(row * self.V), p, C.byref(nmf)))
        return int(nmf.value)

    def reconstruct(self, Xd, Rd, row=0, out_row=0):
        check(self.lib.bm_dbm_reconstruct(self._h, Xd.offset_ptr(row * self.V), Rd.offset_ptr(out_row * self.V)))

    def sample_v(self, k, Vd=None):
        check(self.lib.bm_dbm_sample_v(self._h, k, Vd.ptr if Vd is not None else None))

    def ais(self, n_betas, n_runs, k, seed, chain0=0):
        out = np.empty(n_runs, dtype=np.float32)
        check(self.lib.bm_dbm_ais(self._h, n_betas, n_runs, k, int(seed), int(chain0), out.ctypes.data_as(C.c_void_p)))
        return out

    def log_proba(self, Xd, row=0):
        out = np.empty(self.N, dtype=np.float32)
        check(self.lib.bm_dbm_log_proba(self._h, Xd.offset_ptr(row * self.V), out.ctypes.data_as(C.c_void_p)))
        return out

    def timer_start(self):
        check(self.lib.bm_dbm_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float()
        check(self.lib.bm_dbm_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)

    def timer_mark(self):
        check(self.lib.bm_dbm_timer_mark(self._h))

    def timer_elapsed(self):
        ms = C.c_float()
        check(self.lib.bm_dbm_timer_elapsed(self._h, C.byref(ms)))
        return float(ms.value)



class DbmEngine64(DbmEngine):
    """float64 DBM handle (bm_dbm64_*, include/bm355.h): the method names of DbmEngine with float64 device arrays and
    scalars.  Compatibility path for DBM(dtype='float64') (base/mixin.py:14-25): Bernoulli hidden layers, one process."""

    dtype = np.float64

    def _create(self, cfg, hyper):
        mf_tol, l2, max_norm, damping, st, sc = hyper
        if not np.isfinite(max_norm):
            max_norm = float(np.finfo(np.float64).max)
        pad = lambda v: list(v) + [0.0] * (4 - len(v))
        h12 = (C.c_double * 12)(mf_tol, l2, max_norm, damping, *(pad(st) + pad(sc)))
        check(self.lib.bm_dbm64_create(C.byref(cfg), h12, C.byref(self._h)))

    def set(self, name, value):
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=np.float64), self.shape(name)))
        check(self.lib.bm_dbm64_set_param(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))

    def get(self, name):
        a = np.empty(self.shape(name), dtype=np.float64)
        check(self.lib.bm_dbm64_get_param(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))
        return a

    def seed(self, seed):
        check(self.lib.bm_dbm64_seed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF))

    def set_row_offset(self, row0, particle0):
        check(self.lib.bm_dbm64_set_row_offset(self._h, int(row0), int(particle0)))

    def sync(self):
        check(self.lib.bm_dbm64_sync(self._h))

    def train_step(self, Xd, lr, momentum, k, row=0, want_msre=False):
        nmf, msre = C.c_int32(), C.c_double()
        check(self.lib.bm_dbm64_train_step(self._h, Xd.offset_ptr(row * self.V), lr, momentum, k, C.byref(nmf),
                                           C.byref(msre) if want_msre else None))
        return int(nmf.value), (float(msre.value) if want_msre else None)

    def metrics(self, Xd, k, row=0):
        nmf, msre = C.c_int32(), C.c_double()
        check(self.lib.bm_dbm64_metrics(self._h, Xd.offset_ptr(row * self.V), k, C.byref(nmf), C.byref(msre)))
        return int(nmf.value), float(msre.value)

    def mean_field(self, Xd, row=0, out=None, out_row=0):
        nmf = C.c_int32()
        p = out.offset_ptr(out_row * self.n_hiddens[-1]) if out is not None else None
        check(self.lib.bm_dbm64_mean_field(self._h, Xd.offset_ptr(row * self.V), p, C.byref(nmf)))
        return int(nmf.value)

    def reconstruct(self, Xd, Rd, row=0, out_row=0):
        check(self.lib.bm_dbm64_reconstruct(self._h, Xd.offset_ptr(row * self.V), Rd.offset_ptr(out_row * self.V)))

    def sample_v(self, k, Vd=None):
        check(self.lib.bm_dbm64_sample_v(self._h, k, Vd.ptr if Vd is not None else None))

    def ais(self, n_betas, n_runs, k, seed, chain0=0):
        out = np.empty(n_runs, dtype=np.float64)
        check(self.lib.bm_dbm64_ais(self._h, n_betas, n_runs, k, int(seed) & 0xFFFFFFFFFFFFFFFF, int(chain0),
                                    out.ctypes.data_as(C.c_void_p)))
        return out

    def log_proba(self, Xd, row=0):
        out = np.empty(self.N, dtype=np.float64)
        check(self.lib.bm_dbm64_log_proba(self._h, Xd.offset_ptr(row * self.V), out.ctypes.data_as(C.c_void_p)))
        return out

    def _unsupported(self, *a, **kw):
        raise NotImplementedError('the float64 DBM path is a single-process compatibility path (csrc/bm_dbm64.hip)')

    grad_step = apply_step = set_comm = set_xchg = ais_sharded = ais_sharded_direct = set_mf_allreduce = _unsupported
    device_view = stream = timer_start = timer_stop = timer_mark = timer_elapsed = _unsupported

    def set_fast_binary(self, on, everywhere=False):
        if on:
            self._unsupported()

    def set_ais_literal(self, on):
        pass                                    # float64 AIS accumulates in double: the model dtype IS the literal order's dtype

    def set_sigmoid_literal(self, on):
        if on:
            raise ValueError('the literal float32 tf.sigmoid is a float32 notion')
