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
                ('sp_damping', C.c_float), ('dropout', C.c_float),
                ('h_unit', C.c_int32), ('n_samples', C.c_int32)]


class RbmState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('W', 'vb', 'hb', 'dW', 'dvb', 'dhb', 'q', 'sigma')]


class RbmWork(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('Xin', 'h0m', 'h0s', 'vm', 'vs', 'hm', 'hs')]


def _stale():
    srcs = [os.path.join(HERE, f) for f in ('bm_oracle.c', 'bm_oracle_dbm64.c')]
    return not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in srcs)


def build(force=False):
    if force or _stale():
        import fcntl
        with open(os.path.join(HERE, '.build.lock'), 'w') as lock:      # several ranks may import at once
            fcntl.flock(lock, fcntl.LOCK_EX)
            if force or _stale():
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
        for f in (L.orc_sigmoid_literal, L.orc_exp_eigen):
            f.restype, f.argtypes = C.c_float, [C.c_float]
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
        L.orc_rbm_free_energy_ex.restype = C.c_double
        L.orc_rbm_free_energy_ex.argtypes = [C.POINTER(RbmCfg), C.POINTER(RbmState), f32p, C.c_int, C.c_void_p,
                                             C.c_uint64, C.c_uint32, C.c_uint32]
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
                 sparsity_damping=0.9, dropout=None, h_unit=0, n_samples=0):
        self.V, self.H = int(n_visible), int(n_hidden)
        self.cfg = RbmCfg(self.V, self.H, int(v_unit), int(bool(sample_v_states)), int(bool(sample_h_states)),
                          int(bool(dbm_first)), int(bool(dbm_last)), l2, sparsity_target, sparsity_cost,
                          sparsity_damping, -1.0 if dropout is None else float(dropout), int(h_unit), int(n_samples))
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
        dropped = self.cfg.dropout >= 0.0
        if dropped:
            # free_energy_op reads self._X_batch AFTER tf.nn.dropout replaced it (base_rbm.py:417-418, :516)
            keep = np.float32(self.cfg.dropout)
            u = uniform(self.seed, 1, self.call, X.size, idx0=self.row0 * self.V).reshape(X.shape)
            X = np.ascontiguousarray((X / keep) * np.floor(keep + u), dtype=np.float32)
        fe = lib().orc_rbm_free_energy_ex(C.byref(self.cfg), C.byref(self._state()), X, len(X), None,
                                          self.seed, self.call, 0)
        if self.cfg.h_unit == 2 or dropped:   # the random h_hat / dropout mask consumes one call of the stream
            self.call += 1
        return fe

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
                16 + self.cfg.n_samples if self.cfg.h_unit == 2 else 0, self.cfg.sample_h, None, _ptr(Hn),
                self.seed, 4 + 16 * t, self.call, self.row0)
            Hs = Hn
        self.call += 1
        return Hs, Vs


f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags='C_CONTIGUOUS')


class OracleRBM64(object):
    """float64 twin of a bm_rbm64 handle (oracle/bm_oracle.c, float64 RBM path)."""

    def __init__(self, n_visible, n_hidden, v_unit=0, sample_v_states=False, sample_h_states=True,
                 dbm_first=False, dbm_last=False, l2=1e-4, sparsity_target=0.1, sparsity_cost=0.,
                 sparsity_damping=0.9, dropout=None, h_unit=0, n_samples=0):
        self.V, self.H = int(n_visible), int(n_hidden)
        self.cfg = RbmCfg(self.V, self.H, int(v_unit), int(bool(sample_v_states)), int(bool(sample_h_states)),
                          int(bool(dbm_first)), int(bool(dbm_last)), l2, sparsity_target, sparsity_cost,
                          sparsity_damping, -1.0 if dropout is None else float(dropout), int(h_unit), int(n_samples))
        self.hy = np.array([l2, sparsity_target, sparsity_cost, sparsity_damping,
                            -1.0 if dropout is None else float(dropout)], dtype=np.float64)
        V, H = self.V, self.H
        z = lambda *s: np.zeros(s, dtype=np.float64)
        self.p = dict(W=z(V, H), vb=z(V), hb=z(H), dW=z(V, H), dvb=z(V), dhb=z(H), q_means=z(H), sigma=np.ones(V))
        self.seed = self.call = self.row0 = 0
        self.work = None
        L = lib()
        L.orc_rbm_chain_d.argtypes = [C.POINTER(RbmCfg), f64p, C.POINTER(RbmState), f64p, C.c_int, C.c_int,
                                      C.c_uint64, C.c_uint32, C.c_int64, C.POINTER(RbmWork)]
        L.orc_rbm_train_step_d.argtypes = [C.POINTER(RbmCfg), f64p, C.POINTER(RbmState), f64p, C.c_int, C.c_double,
                                           C.c_double, C.c_int, C.c_uint64, C.c_uint32, C.c_int64, C.POINTER(RbmWork)]
        L.orc_rbm_free_energy_d_ex.restype = C.c_double
        L.orc_rbm_free_energy_d_ex.argtypes = [C.POINTER(RbmCfg), C.POINTER(RbmState), f64p, C.c_int, C.c_void_p,
                                               C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_rbm_metrics_d.argtypes = [C.POINTER(RbmCfg), f64p, C.POINTER(RbmState), C.POINTER(RbmWork), C.c_int,
                                        C.c_uint64, C.c_uint32, C.c_int64, f64p]
        L.orc_sigmoid_d.restype = C.c_double
        L.orc_sigmoid_d.argtypes = [C.c_double]

    def set_seed(self, seed):
        self.seed, self.call = int(seed), 0

    def _state(self):
        p = self.p
        return RbmState(*[_ptr(p[n]) for n in ('W', 'vb', 'hb', 'dW', 'dvb', 'dhb', 'q_means', 'sigma')])

    def _work(self, B):
        V, H = self.V, self.H
        z = lambda *s: np.zeros(s, dtype=np.float64)
        self.work = dict(Xin=z(B, V), h0m=z(B, H), h0s=z(B, H), vm=z(B, V), vs=z(B, V), hm=z(B, H), hs=z(B, H))
        return RbmWork(*[_ptr(self.work[n]) for n in ('Xin', 'h0m', 'h0s', 'vm', 'vs', 'hm', 'hs')])

    def chain(self, X, k):
        X = np.ascontiguousarray(X, dtype=np.float64)
        w = self._work(len(X))
        lib().orc_rbm_chain_d(C.byref(self.cfg), self.hy, C.byref(self._state()), X, len(X), k,
                              self.seed, self.call, self.row0, C.byref(w))
        return w

    def train_step(self, X, lr, momentum, k):
        X = np.ascontiguousarray(X, dtype=np.float64)
        w = self._work(len(X))
        lib().orc_rbm_train_step_d(C.byref(self.cfg), self.hy, C.byref(self._state()), X, len(X), lr, momentum, k,
                                   self.seed, self.call, self.row0, C.byref(w))
        self.call += 1

    def transform(self, X, k):
        self.chain(X, k)
        self.call += 1
        return self.work['hm'].copy()

    def metrics(self, X, k):
        w = self.chain(X, k)
        out = np.zeros(4, dtype=np.float64)
        lib().orc_rbm_metrics_d(C.byref(self.cfg), self.hy, C.byref(self._state()), C.byref(w), len(X),
                                self.seed, self.call, self.row0, out)
        self.call += 1
        return out

    def free_energy(self, X):
        X = np.ascontiguousarray(X, dtype=np.float64)
        if self.cfg.v_unit == 1:
            X = np.ascontiguousarray(X / self.p['sigma'][None, :])
        dropped = self.hy[4] >= 0.0
        if dropped:                           # base_rbm.py:417-418, :516 (see OracleRBM.free_energy)
            keep = self.hy[4]
            u = uniform_d(self.seed, 1, self.call, X.size, idx0=self.row0 * self.V).reshape(X.shape)
            X = np.ascontiguousarray((X / keep) * np.floor(keep + u))
        fe = lib().orc_rbm_free_energy_d_ex(C.byref(self.cfg), C.byref(self._state()), X, len(X), None,
                                            self.seed, self.call, 0)
        if dropped or self.cfg.h_unit == 2:   # the dropout mask / the random h_hat consumed one call of the stream
            self.call += 1
        return fe


def uniform_d(seed, site, call, n, idx0=0):
    out = np.zeros(n, dtype=np.float64)
    f = lib().orc_uniform_d
    f.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, f64p]
    f(seed, site, call, idx0, n, out)
    return out


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


# ------------------------------------------------------------------------------ DBM
MAXL = 4


class DbmCfg(C.Structure):
    _fields_ = [('L', C.c_int32), ('V', C.c_int32), ('n', C.c_int32 * MAXL),
                ('v_unit', C.c_int32), ('sample_v', C.c_int32), ('sample_h', C.c_int32 * MAXL),
                ('N', C.c_int32), ('M', C.c_int32), ('max_mf', C.c_int32),
                ('mf_tol', C.c_float), ('l2', C.c_float), ('max_norm', C.c_float),
                ('sp_target', C.c_float * MAXL), ('sp_cost', C.c_float * MAXL), ('sp_damping', C.c_float),
                ('h_unit', C.c_int32 * MAXL), ('n_samples', C.c_int32 * MAXL), ('sigmoid_literal', C.c_int32)]


class DbmCfg64(C.Structure):
    """orc_dbm_cfg_d (oracle/bm_oracle_dbm64.c): the same fields, hyper-parameters in double, no literal-sigmoid switch"""
    _fields_ = [('L', C.c_int32), ('V', C.c_int32), ('n', C.c_int32 * MAXL),
                ('v_unit', C.c_int32), ('sample_v', C.c_int32), ('sample_h', C.c_int32 * MAXL),
                ('N', C.c_int32), ('M', C.c_int32), ('max_mf', C.c_int32),
                ('mf_tol', C.c_double), ('l2', C.c_double), ('max_norm', C.c_double),
                ('sp_target', C.c_double * MAXL), ('sp_cost', C.c_double * MAXL), ('sp_damping', C.c_double),
                ('h_unit', C.c_int32 * MAXL), ('n_samples', C.c_int32 * MAXL)]


class DbmState(C.Structure):
    _fields_ = [(n, C.c_void_p * MAXL) for n in ('W', 'dW', 'hb', 'dhb', 'q', 'mm', 'mu', 'mu_new', 'H', 'H_new')] + \
               [(n, C.c_void_p) for n in ('vb', 'dvb', 'sigma', 'v', 'v_new')] + [('wnorm', C.c_void_p * MAXL)]


def _dbm_lib():
    L = lib()
    if not getattr(L, '_dbm_ready', False):
        cp, sp = C.POINTER(DbmCfg), C.POINTER(DbmState)
        L.orc_dbm_mean_field.restype = C.c_int
        L.orc_dbm_mean_field.argtypes = [cp, sp, f32p]
        L.orc_dbm_particles.argtypes = [cp, sp, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_int64]
        L.orc_dbm_reconstruct_from_mu.argtypes = [cp, sp, f32p]
        L.orc_dbm_train_step.restype = C.c_int
        L.orc_dbm_train_step.argtypes = [cp, sp, f32p, C.c_float, C.c_float, C.c_int, C.c_uint64, C.c_uint32,
                                         C.c_int64, C.POINTER(C.c_float)]
        L.orc_dbm_sample_v.argtypes = [cp, sp, C.c_int, C.c_uint64, C.c_uint32, C.c_int64]
        L.orc_dbm_ais.argtypes = [cp, sp, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int64, f32p]
        L.orc_dbm_ais_literal.argtypes = [cp, sp, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int64, f32p]
        L.orc_dbm_log_proba.argtypes = [cp, sp, f32p, f32p]
        cd = C.POINTER(DbmCfg64)
        L.orc_dbm_mean_field_d.restype = C.c_int
        L.orc_dbm_mean_field_d.argtypes = [cd, sp, f64p]
        L.orc_dbm_particles_d.argtypes = [cd, sp, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_int64]
        L.orc_dbm_reconstruct_from_mu_d.argtypes = [cd, sp, f64p]
        L.orc_dbm_train_step_d.restype = C.c_int
        L.orc_dbm_train_step_d.argtypes = [cd, sp, f64p, C.c_double, C.c_double, C.c_int, C.c_uint64, C.c_uint32,
                                           C.c_int64, C.POINTER(C.c_double)]
        L.orc_dbm_sample_v_d.argtypes = [cd, sp, C.c_int, C.c_uint64, C.c_uint32, C.c_int64]
        L.orc_dbm_ais_d.argtypes = [cd, sp, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int64, f64p]
        L.orc_dbm_log_proba_d.argtypes = [cd, sp, f64p, f64p]
        L._dbm_ready = True
    return L


class OracleDBM(object):
    """CPU twin of one bm_dbm handle (same variable names as DbmEngine.get/set)."""
    REAL, CFG, SFX, CREAL = np.float32, DbmCfg, '', C.c_float

    def __init__(self, n_visible, n_hiddens, v_unit=0, sample_v_states=True, sample_h_states=None,
                 n_particles=100, batch_size=100, max_mf_updates=10, mf_tol=1e-7, l2=0., max_norm=np.inf,
                 sparsity_target=0.1, sparsity_cost=0., sparsity_damping=0.9, h_units=None, n_samples=None,
                 sigmoid_literal=False):
        self.V, self.nh = int(n_visible), [int(x) for x in n_hiddens]
        self.L, self.N, self.M = len(self.nh), int(batch_size), int(n_particles)
        c = self.CFG()
        for i in range(self.L):     # hidden layer kinds: 0 Bernoulli, 2 Multinomial(n_samples[i]) (layers.py:39-70)
            c.h_unit[i] = int((h_units or [0] * self.L)[i])
            c.n_samples[i] = int((n_samples or [0] * self.L)[i])
        c.L, c.V, c.v_unit, c.sample_v = self.L, self.V, int(v_unit), int(bool(sample_v_states))
        sh = sample_h_states or [True] * self.L
        st = sparsity_target if hasattr(sparsity_target, '__iter__') else [sparsity_target] * self.L
        sc = sparsity_cost if hasattr(sparsity_cost, '__iter__') else [sparsity_cost] * self.L
        for i in range(self.L):
            c.n[i], c.sample_h[i], c.sp_target[i], c.sp_cost[i] = self.nh[i], int(bool(sh[i])), st[i], sc[i]
        c.N, c.M, c.max_mf = self.N, self.M, int(max_mf_updates)
        c.mf_tol, c.l2, c.sp_damping = mf_tol, l2, sparsity_damping
        c.max_norm = float(max_norm) if np.isfinite(max_norm) or self.REAL is np.float32 else float(np.finfo(np.float64).max)
        if self.SFX == '':
            c.sigmoid_literal = int(bool(sigmoid_literal))
        self.cfg = c
        n = [self.V] + self.nh
        z = lambda *s: np.zeros(s, dtype=self.REAL)
        self.p = dict(vb=z(self.V), dvb=z(self.V), sigma=np.ones(self.V, dtype=self.REAL),
                      v=z(self.M, self.V), v_new=z(self.M, self.V))
        for i in range(self.L):
            sfx = '' if i == 0 else '_%d' % i
            self.p['W' + sfx] = z(n[i], n[i + 1]); self.p['dW' + sfx] = z(n[i], n[i + 1])
            for nm in ('hb', 'dhb', 'q_means', 'mu_means', 'W_norm'):
                self.p[nm + sfx] = z(n[i + 1])
            for nm in ('mu', 'mu_new'):
                self.p[nm + sfx] = z(self.N, n[i + 1])
            for nm in ('h', 'h_new'):
                self.p[nm + sfx] = z(self.M, n[i + 1])
        self.seed, self.call, self.prow0 = 0, 0, 0

    def set_seed(self, seed):
        self.seed, self.call = int(seed), 0

    def set_sigmoid_literal(self, on):
        """Bernoulli activations as the reference's float32 tf.sigmoid (orc_sigmoid_literal) - bm_dbm_set_sigmoid_literal"""
        self.cfg.sigmoid_literal = int(bool(on))

    def _state(self):
        s = DbmState()
        names = dict(W='W', dW='dW', hb='hb', dhb='dhb', q='q_means', mm='mu_means', mu='mu', mu_new='mu_new',
                     H='h', H_new='h_new', wnorm='W_norm')
        for fld, nm in names.items():
            arr = getattr(s, fld)
            for i in range(self.L):
                arr[i] = _ptr(self.p[nm + ('' if i == 0 else '_%d' % i)])
        for nm in ('vb', 'dvb', 'sigma', 'v', 'v_new'):
            setattr(s, nm, _ptr(self.p[nm]))
        return s

    def _sync_back(self, s):
        """the C side swaps particle pointers: re-bind names to the buffers that now hold them"""
        byaddr = {a.ctypes.data: a for a in self.p.values()}
        self.p['v'], self.p['v_new'] = byaddr[s.v], byaddr[s.v_new]
        for i in range(self.L):
            sfx = '' if i == 0 else '_%d' % i
            self.p['h' + sfx], self.p['h_new' + sfx] = byaddr[s.H[i]], byaddr[s.H_new[i]]

    def _f(self, name):
        return getattr(_dbm_lib(), name + self.SFX)

    def _x(self, X):
        return np.ascontiguousarray(X, dtype=self.REAL)

    def mean_field(self, X):
        s = self._state()
        n = self._f('orc_dbm_mean_field')(C.byref(self.cfg), C.byref(s), self._x(X))
        self.call += 1
        return n

    def train_step(self, X, lr, momentum, k, want_msre=False):
        s = self._state()
        msre = self.CREAL()
        n = self._f('orc_dbm_train_step')(C.byref(self.cfg), C.byref(s), self._x(X),
                                          lr, momentum, k, self.seed, self.call, self.prow0,
                                          C.byref(msre) if want_msre else None)
        self._sync_back(s)
        self.call += 1
        return n, (msre.value if want_msre else None)

    def metrics(self, X, k):
        """validation fetch (dbm.py:813 under the control dependencies of :521-523): mean-field, k PCD
        sweeps on the particles, reconstruction msre; no parameter update."""
        s = self._state()
        X = self._x(X)
        n = self._f('orc_dbm_mean_field')(C.byref(self.cfg), C.byref(s), X)
        self._f('orc_dbm_particles')(C.byref(self.cfg), C.byref(s), k, 1, self.seed, self.call, self.prow0)
        R = np.zeros((self.N, self.V), dtype=self.REAL)
        self._f('orc_dbm_reconstruct_from_mu')(C.byref(self.cfg), C.byref(s), R)
        self._sync_back(s)
        self.call += 1
        return n, float(np.mean((X.astype(np.float64) - R.astype(np.float64)) ** 2))

    def reconstruct(self, X):
        s = self._state()
        self._f('orc_dbm_mean_field')(C.byref(self.cfg), C.byref(s), self._x(X))
        R = np.zeros((self.N, self.V), dtype=self.REAL)
        self._f('orc_dbm_reconstruct_from_mu')(C.byref(self.cfg), C.byref(s), R)
        self.call += 1
        return R

    def sample_v(self, k):
        s = self._state()
        self._f('orc_dbm_sample_v')(C.byref(self.cfg), C.byref(s), k, self.seed, self.call, self.prow0)
        self._sync_back(s)
        self.call += 1
        return self.p['v'].copy()

    def ais(self, n_betas, n_runs, k, seed, chain0=0, literal=False):
        """literal: the reference's float32 accumulation order (dbm.py:708-728); default: double, difference form"""
        s = self._state()
        out = np.zeros(n_runs, dtype=self.REAL)
        f = _dbm_lib().orc_dbm_ais_literal if (literal and self.SFX == '') else self._f('orc_dbm_ais')
        f(C.byref(self.cfg), C.byref(s), n_betas, n_runs, k, int(seed), int(chain0), out)
        return out

    def log_proba(self, X):
        s = self._state()
        out = np.zeros(self.N, dtype=self.REAL)
        self._f('orc_dbm_log_proba')(C.byref(self.cfg), C.byref(s), self._x(X), out)
        self.call += 1
        return out


class OracleDBM64(OracleDBM):
    """the same in IEEE double (oracle/bm_oracle_dbm64.c): the DBM of a reference model built with dtype='float64'
    (base/mixin.py:14-25)"""
    REAL, CFG, SFX, CREAL = np.float64, DbmCfg64, '_d', C.c_double

    def set_sigmoid_literal(self, on):
        if on:
            raise ValueError('the literal float32 tf.sigmoid is a float32 notion')
