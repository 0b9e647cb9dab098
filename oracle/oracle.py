"""ctypes front-end of the C oracle (oracle/bm_oracle.c) + small numpy helpers.

TEST INFRASTRUCTURE ONLY (see the header of bm_oracle.c): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'libbm_oracle.so')

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags='C_CONTIGUOUS')
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags='C_CONTIGUOUS')
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags='C_CONTIGUOUS')


class RbmCfg(C.Structure):
    _fields_ = [('V', C.c_int32), ('H', C.c_int32),
                ('v_unit', C.c_int32), ('sample_v', C.c_int32), ('sample_h', C.c_int32),
                ('dbm_first', C.c_int32), ('dbm_last', C.c_int32),
                ('l2', C.c_float), ('sp_target', C.c_float), ('sp_cost', C.c_float),
                ('sp_damping', C.c_float), ('dropout', C.c_float)]


class RbmState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('W', 'vb', 'hb', 'dW', 'dvb', 'dhb', 'q', 'sigma')]


class RbmWork(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('Xin', 'h0m', 'h0s', 'vm', 'vs', 'hm', 'hs')]


def build(force=False):
    src = os.path.join(HERE, 'bm_oracle.c')
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', HERE, '-s', 'clean', 'all'])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.orc_sigmoid.restype = C.c_float
        L.orc_sigmoid.argtypes = [C.c_float]
        L.orc_philox_words.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, u32p]
        L.orc_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, f32p]
        L.orc_normal.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, f32p]
        L.orc_rbm_chain.argtypes = [C.POINTER(RbmCfg), C.POINTER(RbmState), f32p, C.c_int, C.c_int,
                                    C.c_uint64, C.c_uint32, C.c_int64, C.POINTER(RbmWork)]
        L.orc_rbm_raw_grads.argtypes = [C.POINTER(RbmCfg), C.POINTER(RbmWork), C.c_int, f32p]
        L.orc_rbm_apply.argtypes = [C.POINTER(RbmCfg), C.POINTER(RbmState), f32p, C.c_float, C.c_float, C.c_float]
        L.orc_rbm_train_step.argtypes = [C.POINTER(RbmCfg), C.POINTER(RbmState), f32p, C.c_int, C.c_float,
                                         C.c_float, C.c_int, C.c_uint64, C.c_uint32, C.c_int64,
                                         C.POINTER(RbmWork)]
        L.orc_rbm_free_energy.restype = C.c_double
        L.orc_rbm_free_energy.argtypes = [C.POINTER(RbmCfg), C.POINTER(RbmState), f32p, C.c_int, C.c_void_p]
        L.orc_rbm_metrics.argtypes = [C.POINTER(RbmCfg), C.POINTER(RbmState), C.POINTER(RbmWork), C.c_int,
                                      C.c_uint64, C.c_uint32, C.c_int64, f32p, i32p]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleRBM(object):
    """CPU twin of one bm_rbm handle: same state names, same call sequence."""

    def __init__(self, n_visible, n_hidden, v_unit=0, sample_v_states=False, sample_h_states=True,
                 dbm_first=False, dbm_last=False, l2=1e-4, sparsity_target=0.1, sparsity_cost=0.,
                 sparsity_damping=0.9, dropout=None):
        self.V, self.H = int(n_visible), int(n_hidden)
        self.cfg = RbmCfg(self.V, self.H, int(v_unit), int(bool(sample_v_states)), int(bool(sample_h_states)),
                          int(bool(dbm_first)), int(bool(dbm_last)), l2, sparsity_target, sparsity_cost,
                          sparsity_damping, -1.0 if dropout is None else float(dropout))
        V, H = self.V, self.H
        z = lambda *s: np.zeros(s, dtype=np.float32)
        self.p = dict(W=z(V, H), vb=z(V), hb=z(H), dW=z(V, H), dvb=z(V), dhb=z(H), q_means=z(H),
                      sigma=np.ones(V, dtype=np.float32))
        self.seed = 0
        self.call = 0
        self.row0 = 0
        self.work = None

    def set_seed(self, seed):
        self.seed, self.call = int(seed), 0

    def _state(self):
        p = self.p
        return RbmState(_ptr(p['W']), _ptr(p['vb']), _ptr(p['hb']), _ptr(p['dW']), _ptr(p['dvb']),
                        _ptr(p['dhb']), _ptr(p['q_means']), _ptr(p['sigma']))

    def _work(self, B):
        V, H = self.V, self.H
        z = lambda *s: np.zeros(s, dtype=np.float32)
        self.work = dict(Xin=z(B, V), h0m=z(B, H), h0s=z(B, H), vm=z(B, V), vs=z(B, V), hm=z(B, H), hs=z(B, H))
        w = self.work
        return RbmWork(*[_ptr(w[n]) for n in ('Xin', 'h0m', 'h0s', 'vm', 'vs', 'hm', 'hs')])

    def chain(self, X, k):
        X = np.ascontiguousarray(X, dtype=np.float32)
        w = self._work(len(X))
        lib().orc_rbm_chain(C.byref(self.cfg), C.byref(self._state()), X, len(X), k,
                            self.seed, self.call, self.row0, C.byref(w))
        return w

    def train_step(self, X, lr, momentum, k):
        X = np.ascontiguousarray(X, dtype=np.float32)
        w = self._work(len(X))
        lib().orc_rbm_train_step(C.byref(self.cfg), C.byref(self._state()), X, len(X), lr, momentum, k,
                                 self.seed, self.call, self.row0, C.byref(w))
        self.call += 1

    def raw_grads(self, X, k):
        """phase 1 of the data-parallel split: chain + raw sums (no update)."""
        w = self.chain(X, k)
        raw = np.zeros(self.V * self.H + self.V + 2 * self.H, dtype=np.float32)
        lib().orc_rbm_raw_grads(C.byref(self.cfg), C.byref(w), len(X), raw)
        self.call += 1
        return raw

    def apply(self, raw, N, lr, momentum):
        lib().orc_rbm_apply(C.byref(self.cfg), C.byref(self._state()), np.ascontiguousarray(raw), N, lr, momentum)

    def transform(self, X, k):
        self.chain(X, k)
        self.call += 1
        return self.work['hm'].copy()

    def metrics(self, X, k, advance=True):
        w = self.chain(X, k)
        out = np.zeros(4, dtype=np.float32)
        flip = np.zeros(len(X), dtype=np.int32)
        lib().orc_rbm_metrics(C.byref(self.cfg), C.byref(self._state()), C.byref(w), len(X),
                              self.seed, self.call, self.row0, out, flip)
        if advance:
            self.call += 1
        return out, flip

    def free_energy(self, X):
        X = np.ascontiguousarray(X, dtype=np.float32)
        if self.cfg.v_unit == 1:
            X = np.ascontiguousarray(X / self.p['sigma'][None, :], dtype=np.float32)
        return lib().orc_rbm_free_energy(C.byref(self.cfg), C.byref(self._state()), X, len(X), None)

    def gibbs(self, Hs, n_steps):
        """pure sampling sweep: n_steps of h->v->h from hidden states (bm_rbm_gibbs)."""
        L = lib()
        V, H = self.V, self.H
        Hs = np.ascontiguousarray(Hs, dtype=np.float32).copy()
        B = len(Hs)
        Vs = np.zeros((B, V), dtype=np.float32)
        Wt = np.ascontiguousarray(self.p['W'].T)
        up = 1.0 + float(self.cfg.dbm_first)
        down = 1.0 + float(self.cfg.dbm_last)
        act = L.orc_act
        act.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                        C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                        C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64]
        for t in range(n_steps):
            act(_ptr(Hs), H, _ptr(Wt), None, 0, None, V, B, _ptr(self.p['vb']), _ptr(self.p['sigma']), down,
                self.cfg.v_unit, self.cfg.sample_v, None, _ptr(Vs), self.seed, 3 + 16 * t, self.call, self.row0)
            Hn = np.zeros_like(Hs)
            act(_ptr(Vs), V, _ptr(self.p['W']), None, 0, None, H, B, _ptr(self.p['hb']), None, up,
                0, self.cfg.sample_h, None, _ptr(Hn), self.seed, 4 + 16 * t, self.call, self.row0)
            Hs = Hn
        self.call += 1
        return Hs, Vs


def philox_words(seed, site, call, block0, nblocks):
    out = np.zeros(4 * nblocks, dtype=np.uint32)
    lib().orc_philox_words(seed, site, call, block0, nblocks, out)
    return out.reshape(nblocks, 4)


def uniform(seed, site, call, n, idx0=0):
    out = np.zeros(n, dtype=np.float32)
    lib().orc_uniform(seed, site, call, idx0, n, out)
    return out


def normal(seed, site, call, n, idx0=0):
    out = np.zeros(n, dtype=np.float32)
    lib().orc_normal(seed, site, call, idx0, n, out)
    return out
