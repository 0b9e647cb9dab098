"""ctypes binding of libbm355.so (include/bm355.h).

The HIP library IS the product: there is no CPU/PyTorch fallback.  `load()`
raises if the shared object cannot be built/loaded, and every model call
raises `Bm355Error` if the library reports an error (e.g. no GPU visible).
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

MAX_LAYERS = 4
UNIT_BERNOULLI, UNIT_GAUSSIAN, UNIT_MULTINOMIAL = 0, 1, 2


class Bm355Error(RuntimeError):
    pass


class RbmConfig(C.Structure):
    _fields_ = [('n_visible', C.c_int32), ('n_hidden', C.c_int32), ('v_unit', C.c_int32),
                ('sample_v_states', C.c_int32), ('sample_h_states', C.c_int32),
                ('dbm_first', C.c_int32), ('dbm_last', C.c_int32), ('max_batch', C.c_int32),
                ('l2', C.c_float), ('sparsity_target', C.c_float), ('sparsity_cost', C.c_float),
                ('sparsity_damping', C.c_float), ('dropout', C.c_float),
                ('h_unit', C.c_int32), ('n_samples', C.c_int32)]


class DbmConfig(C.Structure):
    _fields_ = [('n_layers', C.c_int32), ('n_visible', C.c_int32),
                ('n_hiddens', C.c_int32 * MAX_LAYERS), ('v_unit', C.c_int32),
                ('sample_v_states', C.c_int32), ('sample_h_states', C.c_int32 * MAX_LAYERS),
                ('n_particles', C.c_int32), ('batch_size', C.c_int32), ('max_mf_updates', C.c_int32),
                ('mf_tol', C.c_float), ('l2', C.c_float), ('max_norm', C.c_float),
                ('sparsity_target', C.c_float * MAX_LAYERS), ('sparsity_cost', C.c_float * MAX_LAYERS),
                ('sparsity_damping', C.c_float),
                ('h_unit', C.c_int32 * MAX_LAYERS), ('n_samples', C.c_int32 * MAX_LAYERS)]


_vp, _i32, _i64, _u64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_size_t
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)

# name -> argtypes; every function returns int (0 = ok) unless listed in _RESTYPE
SIGNATURES = {
    'bm_device_count': [],
    'bm_set_device': [C.c_int],
    'bm_dev_alloc': [_sz, C.POINTER(_vp)],
    'bm_dev_free': [_vp],
    'bm_h2d': [_vp, _vp, _sz],
    'bm_d2h': [_vp, _vp, _sz],
    'bm_dev_memset': [_vp, C.c_int, _sz],
    'bm_debug_tile_map': [_i32, _i32, C.c_double, C.c_double, _ip, _ip, _ip],
    'bm_comm_unique_id': [_vp],
    'bm_comm_init': [C.c_int32, C.c_int32, _vp, C.POINTER(_vp)],
    'bm_comm_destroy': [_vp],
    'bm_comm_allreduce_sum': [_vp, _vp, _sz, _vp],
    'bm_comm_allgather': [_vp, _vp, _vp, _sz, _vp],
    'bm_comm_allreduce_max': [_vp, _vp, _sz, _vp],
    'bm_comm_rank': [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    'bm_rbm_allreduce_grads': [_vp, _vp],
    'bm_rbm_set_grad_slot': [_vp, _i32],
    'bm_rbm_allreduce_grads_async': [_vp, _vp],
    'bm_rbm_wait_grads': [_vp, _i32],
    'bm_dbm_allreduce_grads': [_vp, _vp],
    'bm_xchg_create': [C.c_int32, C.c_int32, _vp, _sz, C.POINTER(_vp)],
    'bm_xchg_blob_bytes': [],
    'bm_xchg_export': [_vp, _vp],
    'bm_xchg_attach': [_vp, _vp],
    'bm_xchg_destroy': [_vp],
    'bm_xchg_allreduce_sum': [_vp, _vp],
    'bm_xchg_allreduce_max1': [_vp, _vp, _vp],
    'bm_xchg_status': [_vp, C.POINTER(C.c_int32)],
    'bm_xchg_info': [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(_sz)],
    'bm_rbm_xchg_create': [_vp, C.c_int32, C.c_int32, C.POINTER(_vp)],
    'bm_dbm_xchg_create': [_vp, C.c_int32, C.c_int32, C.POINTER(_vp)],
    'bm_rbm_allreduce_grads_direct': [_vp, _vp],
    'bm_dbm_allreduce_grads_direct': [_vp, _vp],
    'bm_rbm_exchange_apply_direct': [_vp, _vp, _i32, C.c_float, C.c_float],
    'bm_rbm_exchange_gather_dw': [_vp, _vp],
    'bm_dbm_exchange_apply_ok': [_vp, _vp, C.POINTER(_i32)],
    'bm_dbm_exchange_apply_direct': [_vp, _vp, _i32, _i32, C.c_float, C.c_float],
    'bm_dbm_exchange_gather_dw': [_vp, _vp],
    'bm_xchg_set_timeout': [_vp, C.c_double],
    'bm_xchg_set_max_workgroups': [_vp, _i32],
    'bm_dbm_set_xchg': [_vp, _vp],
    'bm_dbm_set_fast_binary': [_vp, _i32],
    'bm_dbm_set_ais_literal': [_vp, _i32],
    'bm_dbm_set_sigmoid_literal': [_vp, _i32],
    'bm_rbm_set_fast_binary': [_vp, _i32],
    'bm_rbm64_create': [C.POINTER(RbmConfig), C.POINTER(C.c_double), C.POINTER(_vp)],
    'bm_rbm64_destroy': [_vp],
    'bm_rbm64_sync': [_vp],
    'bm_rbm64_seed': [_vp, C.c_uint64],
    'bm_rbm64_set_row_offset': [_vp, C.c_int64],
    'bm_rbm64_set_param': [_vp, C.c_char_p, _vp, _sz],
    'bm_rbm64_get_param': [_vp, C.c_char_p, _vp, _sz],
    'bm_rbm64_train_step': [_vp, _vp, C.c_int32, C.c_double, C.c_double, C.c_int32],
    'bm_rbm64_train_step_metrics': [_vp, _vp, C.c_int32, C.c_double, C.c_double, C.c_int32, C.POINTER(C.c_double)],
    'bm_rbm64_transform': [_vp, _vp, C.c_int32, C.c_int32, _vp],
    'bm_rbm64_metrics': [_vp, _vp, C.c_int32, C.c_int32, C.POINTER(C.c_double)],
    'bm_rbm64_free_energy': [_vp, _vp, C.c_int32, C.POINTER(C.c_double)],
    'bm_dbm64_create': [C.POINTER(DbmConfig), C.POINTER(C.c_double), C.POINTER(_vp)],
    'bm_dbm64_destroy': [_vp],
    'bm_dbm64_sync': [_vp],
    'bm_dbm64_seed': [_vp, C.c_uint64],
    'bm_dbm64_set_row_offset': [_vp, C.c_int64, C.c_int64],
    'bm_dbm64_set_param': [_vp, C.c_char_p, _vp, _sz],
    'bm_dbm64_get_param': [_vp, C.c_char_p, _vp, _sz],
    'bm_dbm64_train_step': [_vp, _vp, C.c_double, C.c_double, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double)],
    'bm_dbm64_metrics': [_vp, _vp, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double)],
    'bm_dbm64_mean_field': [_vp, _vp, _vp, C.POINTER(C.c_int32)],
    'bm_dbm64_reconstruct': [_vp, _vp, _vp],
    'bm_dbm64_sample_v': [_vp, C.c_int32, _vp],
    'bm_dbm64_ais': [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int64, _vp],
    'bm_dbm64_log_proba': [_vp, _vp, _vp],
    'bm_rbm_create': [C.POINTER(RbmConfig), C.POINTER(_vp)],
    'bm_rbm_destroy': [_vp],
    'bm_rbm_sync': [_vp],
    'bm_rbm_set_param': [_vp, C.c_char_p, _vp, _sz],
    'bm_rbm_get_param': [_vp, C.c_char_p, _vp, _sz],
    'bm_rbm_set_param_dev': [_vp, C.c_char_p, _vp, _sz],
    'bm_rbm_dev_ptr': [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(_sz)],
    'bm_rbm_seed': [_vp, _u64],
    'bm_rbm_set_row_offset': [_vp, _i64],
    'bm_rbm_train_step': [_vp, _vp, _i32, _f32, _f32, _i32],
    'bm_rbm_train_step_metrics': [_vp, _vp, _i32, _f32, _f32, _i32, _fp],
    'bm_rbm_train_step_metrics_async': [_vp, _vp, _i32, _f32, _f32, _i32],
    'bm_rbm_collect_metrics': [_vp, _fp, _i32, C.POINTER(C.c_int32)],
    'bm_rbm_train_epoch': [_vp, _vp, _i64, _i32, _f32, _f32, _i32],
    'bm_rbm_stage': [_vp, _i32],
    'bm_rbm_get_staged': [_vp, _i32, C.c_char_p, _vp, _sz],
    'bm_rbm_grad_step': [_vp, _vp, _i32, _i32],
    'bm_rbm_apply_step': [_vp, _i32, _f32, _f32],
    'bm_rbm_transform': [_vp, _vp, _i32, _i32, _vp],
    'bm_rbm_metrics': [_vp, _vp, _i32, _i32, _fp],
    'bm_rbm_free_energy': [_vp, _vp, _i32, _fp],
    'bm_rbm_gibbs': [_vp, _vp, _vp, _i32, _i32],
    'bm_rbm_stream': [_vp, C.POINTER(_vp)],
    'bm_rbm_profile': [_vp, _i32],
    'bm_rbm_kernel_times': [_vp, _fp, _ip],
    'bm_rbm_chain_stats': [_vp, C.POINTER(C.c_int64)],
    'bm_rbm_timer_start': [_vp],
    'bm_rbm_timer_stop': [_vp, _fp],
    'bm_rbm_timer_mark': [_vp],
    'bm_rbm_timer_elapsed': [_vp, _fp],
    'bm_dbm_create': [C.POINTER(DbmConfig), C.POINTER(_vp)],
    'bm_dbm_destroy': [_vp],
    'bm_dbm_sync': [_vp],
    'bm_dbm_seed': [_vp, _u64],
    'bm_dbm_set_row_offset': [_vp, _i64, _i64],
    'bm_dbm_set_param': [_vp, C.c_char_p, _vp, _sz],
    'bm_dbm_get_param': [_vp, C.c_char_p, _vp, _sz],
    'bm_dbm_dev_ptr': [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(_sz)],
    'bm_dbm_train_step': [_vp, _vp, _f32, _f32, _i32, _ip, _fp],
    'bm_dbm_metrics': [_vp, _vp, _i32, _ip, _fp],
    'bm_dbm_grad_step': [_vp, _vp, _i32, _ip],
    'bm_dbm_apply_step': [_vp, _i32, _i32, _f32, _f32],
    'bm_dbm_set_mf_allreduce': [_vp, _vp, _vp],
    'bm_dbm_set_comm': [_vp, _vp],
    'bm_dbm_ais_sharded': [_vp, _vp, _i32, _i32, _i32, _u64, _vp],
    'bm_dbm_ais_sharded_direct': [_vp, _vp, _i32, _i32, _i32, _u64, _vp],
    'bm_dbm_stream': [_vp, C.POINTER(_vp)],
    'bm_dbm_mean_field': [_vp, _vp, _vp, _ip],
    'bm_dbm_reconstruct': [_vp, _vp, _vp],
    'bm_dbm_sample_v': [_vp, _i32, _vp],
    'bm_dbm_ais': [_vp, _i32, _i32, _i32, _u64, _i64, _vp],
    'bm_dbm_log_proba': [_vp, _vp, _vp],
    'bm_dbm_timer_start': [_vp],
    'bm_dbm_timer_stop': [_vp, _fp],
    'bm_dbm_timer_mark': [_vp],
    'bm_dbm_timer_elapsed': [_vp, _fp],
}
_RESTYPE = {'bm_last_error': C.c_char_p, 'bm_version': C.c_char_p}
MF_REDUCE_FN = C.CFUNCTYPE(C.c_float, C.c_float, C.c_void_p)

_lib = None


def load(rebuild=True):
    """Load (building if needed) libbm355.so; raises if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if rebuild and _build.needs_build():
        import shutil
        have_hipcc = bool(shutil.which('hipcc')) or os.path.exists('/opt/rocm/bin/hipcc')
        try:
            _build.build()
        except Exception as e:
            # A present-but-older library is used ONLY on a box without hipcc (nothing can be rebuilt
            # there).  With a compiler, a failed build means the sources are broken: binding the stale
            # library would silently run code that no longer matches them.
            if have_hipcc or not os.path.exists(path):
                raise Bm355Error('libbm355.so could not be built from the current sources: %s' % e)
            import warnings
            warnings.warn('libbm355.so is older than its sources and hipcc is not available; using the existing '
                          'library (%s)' % e)
    if not os.path.exists(path):
        raise Bm355Error('libbm355.so not found at %s (run __graft_entry__.build())' % path)
    lib = C.CDLL(path)
    for name, restype in _RESTYPE.items():
        f = getattr(lib, name)
        f.restype = restype
        f.argtypes = []
    for name, argtypes in SIGNATURES.items():
        f = getattr(lib, name)   # AttributeError here == header/library mismatch
        f.restype = C.c_int
        f.argtypes = argtypes
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise Bm355Error(load().bm_last_error().decode('utf-8', 'replace'))


def exported_symbols():
    return sorted(list(SIGNATURES) + list(_RESTYPE))


class DeviceArray(object):
    """A float32/int32 array in HBM, allocated through the C-ABI helpers
    (no torch involved).  Exposes `__cuda_array_interface__` so that torch can
    wrap it zero-copy for RCCL collectives (torch.as_tensor(arr, device='cuda'))."""

    def __init__(self, shape, dtype=np.float32, ptr=None, owner=None):
        self.shape = tuple(int(s) for s in (shape if hasattr(shape, '__iter__') else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self._own = ptr is None
        self._owner = owner
        if ptr is None:
            p = _vp()
            check(load().bm_dev_alloc(self.nbytes, C.byref(p)))
            ptr = p.value
        self.ptr = int(ptr)

    @classmethod
    def from_numpy(cls, a, dtype=np.float32):
        a = np.ascontiguousarray(a, dtype=dtype)
        d = cls(a.shape, dtype)
        check(load().bm_h2d(d.ptr, a.ctypes.data_as(_vp), a.nbytes))
        return d

    @classmethod
    def from_numpy_reusing(cls, old, a, dtype=np.float32):
        """like from_numpy, but into `old`'s allocation when it is large enough (repeated fit() calls re-upload the
        training set: hipMalloc + hipFree of a few hundred MB cost tens of milliseconds per call)"""
        a = np.ascontiguousarray(a, dtype=dtype)
        if old is None or not old._own or not old.ptr or getattr(old, '_capacity', old.nbytes) < a.nbytes:
            if old is not None:
                old.free()
            d = cls(a.shape, dtype)
            d._capacity = d.nbytes
        else:
            d = old
            d.shape, d.dtype, d.nbytes = tuple(int(x) for x in a.shape), np.dtype(dtype), a.nbytes
        check(load().bm_h2d(d.ptr, a.ctypes.data_as(_vp), a.nbytes))
        return d

    def numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        check(load().bm_d2h(out.ctypes.data_as(_vp), self.ptr, self.nbytes))
        return out

    def offset_ptr(self, n_elems):
        return self.ptr + int(n_elems) * self.dtype.itemsize

    @property
    def __cuda_array_interface__(self):
        return {'shape': self.shape, 'typestr': self.dtype.str, 'data': (self.ptr, False), 'version': 2}

    def free(self):
        if self._own and self.ptr:
            load().bm_dev_free(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
