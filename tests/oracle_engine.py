"""The C oracle behind the engine interface of the model classes - TEST INFRASTRUCTURE.

`boltzmann_machines_amd.{rbm,dbm}` drive a device engine (`engine.RbmEngine / RbmEngine64 / DbmEngine`: one method per
`session.run` fetch site).  `install(monkeypatch)` swaps those three names, `_ffi.DeviceArray` and `as_device` for
host-side stand-ins backed by `oracle.OracleRBM / OracleRBM64 / OracleDBM`, so that the HOST LOGIC of the package -
schedules, seeds and the MT stream, call counters, batching, snapshot / restore, checkpoints, resume - runs on a box
without a GPU and is checked against the fixtures the reference generated (tests/test_reference_fixtures.py).  The
product never imports this module and has no such switch: without it, a missing GPU is an error."""
import numpy as np

from oracle import oracle as orc


class HostArray(object):
    """stand-in for _ffi.DeviceArray: a host ndarray with the same constructor / accessors"""

    def __init__(self, shape, dtype=np.float32, ptr=None, owner=None):
        self.shape = tuple(int(s) for s in (shape if hasattr(shape, '__iter__') else (shape,)))
        self.dtype = np.dtype(dtype)
        self.a = np.zeros(self.shape, dtype=self.dtype)
        self.nbytes = self.a.nbytes
        self.ptr = 1

    @classmethod
    def from_numpy(cls, a, dtype=np.float32):
        a = np.ascontiguousarray(a, dtype=dtype)
        d = cls(a.shape, dtype)
        d.a[...] = a
        return d

    @classmethod
    def from_numpy_reusing(cls, old, a, dtype=np.float32):
        return cls.from_numpy(a, dtype)

    def numpy(self):
        return self.a.copy()

    def free(self):
        pass


def as_device(X, dtype=np.float32):
    return X if isinstance(X, HostArray) else HostArray.from_numpy(np.asarray(X), dtype)


class _OracleRbmBase(object):
    def _common(self, twin, V, H, max_batch):
        self.twin, self.V, self.H, self.max_batch = twin, int(V), int(H), int(max_batch)

    def close(self):
        self.twin = None

    def set(self, name, value):
        p = self.twin.p[name]
        p[...] = np.broadcast_to(np.asarray(value, dtype=p.dtype), p.shape)

    def get(self, name):
        return self.twin.p[name].copy()

    def seed(self, seed):
        self.twin.set_seed(seed)

    def sync(self):
        pass

    def _rows(self, Xd, B, row):
        return Xd.a[row:row + B]

    def train_step(self, Xd, B, lr, momentum, k, row=0):
        self.twin.train_step(self._rows(Xd, B, row), lr, momentum, k)

    def train_epoch(self, Xd, N, batch, lr, momentum, k, row=0):
        for s in range(0, N, batch):
            self.train_step(Xd, min(batch, N - s), lr, momentum, k, row=row + s)

    def transform(self, Xd, B, k, Hd, row=0, out_row=0):
        Hd.a[out_row:out_row + B] = self.twin.transform(self._rows(Xd, B, row), k)

    def free_energy(self, Xd, B, row=0):
        return float(self.twin.free_energy(self._rows(Xd, B, row)))


class OracleRbmEngine(_OracleRbmBase):
    dtype = np.float32

    def __init__(self, n_visible, n_hidden, v_unit=0, sample_v_states=False, sample_h_states=True, dbm_first=False,
                 dbm_last=False, max_batch=10, l2=1e-4, sparsity_target=0.1, sparsity_cost=0., sparsity_damping=0.9,
                 dropout=None, h_unit=0, n_samples=0):
        self._common(orc.OracleRBM(n_visible, n_hidden, v_unit=v_unit, sample_v_states=sample_v_states,
                                   sample_h_states=sample_h_states, dbm_first=dbm_first, dbm_last=dbm_last, l2=l2,
                                   sparsity_target=sparsity_target, sparsity_cost=sparsity_cost,
                                   sparsity_damping=sparsity_damping, dropout=dropout, h_unit=h_unit,
                                   n_samples=n_samples), n_visible, n_hidden, max_batch)

    def metrics(self, Xd, B, k, row=0):
        out, _ = self.twin.metrics(self._rows(Xd, B, row), k)
        return out

    def train_step_metrics(self, Xd, B, lr, momentum, k, row=0):
        # one session.run fetching [metrics..., train_op] (base_rbm.py:554-564): the metrics of the chain the update
        # uses, on the parameters before the update; ONE call of the RNG stream
        X = self._rows(Xd, B, row)
        out, _ = self.twin.metrics(X, k, advance=False)
        self.twin.train_step(X, lr, momentum, k)
        return out

    MAX_PENDING_METRICS = 4096

    def train_step_metrics_async(self, Xd, B, lr, momentum, k, row=0):
        self.__dict__.setdefault('_pending', []).append(self.train_step_metrics(Xd, B, lr, momentum, k, row=row))

    def collect_metrics(self):
        out, self._pending = np.array(self.__dict__.get('_pending', []), dtype=np.float32).reshape(-1, 4), []
        return out


class OracleRbmEngine64(_OracleRbmBase):
    dtype = np.float64

    def __init__(self, n_visible, n_hidden, v_unit=0, sample_v_states=False, sample_h_states=True, dbm_first=False,
                 dbm_last=False, max_batch=10, l2=1e-4, sparsity_target=0.1, sparsity_cost=0., sparsity_damping=0.9,
                 dropout=None, h_unit=0, n_samples=0):
        self._common(orc.OracleRBM64(n_visible, n_hidden, v_unit=v_unit, sample_v_states=sample_v_states,
                                     sample_h_states=sample_h_states, dbm_first=dbm_first, dbm_last=dbm_last, l2=l2,
                                     sparsity_target=sparsity_target, sparsity_cost=sparsity_cost,
                                     sparsity_damping=sparsity_damping, dropout=dropout, h_unit=h_unit,
                                     n_samples=n_samples), n_visible, n_hidden, max_batch)

    def metrics(self, Xd, B, k, row=0):
        return self.twin.metrics(self._rows(Xd, B, row), k)

    def train_step_metrics(self, Xd, B, lr, momentum, k, row=0):
        X = self._rows(Xd, B, row)
        call = self.twin.call
        out = self.twin.metrics(X, k)
        self.twin.call = call
        self.twin.train_step(X, lr, momentum, k)
        return out


class OracleDbmEngine(object):
    dtype, TWIN = np.float32, orc.OracleDBM

    def __init__(self, n_visible, n_hiddens, v_unit=0, sample_v_states=True, sample_h_states=None, n_particles=100,
                 batch_size=100, max_mf_updates=10, mf_tol=1e-7, l2=0., max_norm=np.inf, sparsity_target=0.1,
                 sparsity_cost=0., sparsity_damping=0.9, h_units=None, n_samples=None):
        self.V, self.n_hiddens = int(n_visible), [int(x) for x in n_hiddens]
        self.N, self.M = int(batch_size), int(n_particles)
        self.twin = self.TWIN(n_visible, n_hiddens, v_unit=v_unit, sample_v_states=sample_v_states,
                                  sample_h_states=sample_h_states, n_particles=n_particles, batch_size=batch_size,
                                  max_mf_updates=max_mf_updates, mf_tol=mf_tol, l2=l2, max_norm=max_norm,
                                  sparsity_target=sparsity_target, sparsity_cost=sparsity_cost,
                                  sparsity_damping=sparsity_damping, h_units=h_units, n_samples=n_samples)

    def close(self):
        self.twin = None

    def set(self, name, value):
        p = self.twin.p[name]
        p[...] = np.broadcast_to(np.asarray(value, dtype=p.dtype), p.shape)

    def get(self, name):
        return self.twin.p[name].copy()

    def seed(self, seed):
        self.twin.set_seed(seed)

    def sync(self):
        pass

    def set_fast_binary(self, on):
        raise RuntimeError('the oracle has no fast-binary mode')

    def set_ais_literal(self, on):
        self._ais_literal = bool(on)

    def set_sigmoid_literal(self, on):
        self.twin.set_sigmoid_literal(on)

    def _rows(self, Xd, row):
        return Xd.a[row:row + self.N]

    def train_step(self, Xd, lr, momentum, k, row=0, want_msre=False):
        return self.twin.train_step(self._rows(Xd, row), lr, momentum, k, want_msre=want_msre)

    def metrics(self, Xd, k, row=0):
        return self.twin.metrics(self._rows(Xd, row), k)

    def mean_field(self, Xd, row=0, out=None, out_row=0):
        n = self.twin.mean_field(self._rows(Xd, row))
        if out is not None:
            sfx = '' if len(self.n_hiddens) == 1 else '_%d' % (len(self.n_hiddens) - 1)
            out.a[out_row:out_row + self.N] = self.twin.p['mu' + sfx]
        return n

    def reconstruct(self, Xd, Rd, row=0, out_row=0):
        Rd.a[out_row:out_row + self.N] = self.twin.reconstruct(self._rows(Xd, row))

    def sample_v(self, k, Vd=None):
        v = self.twin.sample_v(k)
        if Vd is not None:
            Vd.a[...] = v

    def ais(self, n_betas, n_runs, k, seed, chain0=0):
        return self.twin.ais(n_betas, n_runs, k, seed, chain0, literal=getattr(self, '_ais_literal', False))

    def log_proba(self, Xd, row=0):
        return self.twin.log_proba(self._rows(Xd, row))


class OracleDbmEngine64(OracleDbmEngine):
    dtype, TWIN = np.float64, orc.OracleDBM64


def install(monkeypatch):
    """the model classes of boltzmann_machines_amd run on the oracle for the duration of a test"""
    from boltzmann_machines_amd import _ffi, rbm, dbm, engine
    monkeypatch.setenv('BM355_STAGED_SAVE', '0')
    monkeypatch.setattr(_ffi, 'DeviceArray', HostArray)
    monkeypatch.setattr(engine, 'as_device', as_device)
    monkeypatch.setattr(dbm, 'as_device', as_device)
    monkeypatch.setattr(rbm, 'RbmEngine', OracleRbmEngine)
    monkeypatch.setattr(rbm, 'RbmEngine64', OracleRbmEngine64)
    monkeypatch.setattr(dbm, 'DbmEngine', OracleDbmEngine)
    monkeypatch.setattr(dbm, 'DbmEngine64', OracleDbmEngine64)
