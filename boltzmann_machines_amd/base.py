"""Model plumbing shared by the RBM and DBM classes.

Mirrors the reference's base/ package with the TensorFlow parts replaced by
the MI355X engine:

* `BaseMixin` / `DtypeMixin` / `SeedMixin`   reference base/mixin.py:7-35
* `BaseModel.get_params/set_params`          reference base/base_model.py:12-63
* `EngineModel` (the reference's `TensorFlowModel`, base/tf_model.py:43-202):
  working paths, params.json / random_state.json persistence, `init`, `fit`,
  `get_tf_params`, `load_model`, and `run_on_engine` — the stand-in for the
  `run_in_tf_session` decorator (tf_model.py:10-40).

Differences that are deliberate (DESIGN.md "boundary"):
  - variables live in HBM behind a C-ABI handle for the lifetime of the Python
    object; they are written to `<model_filepath>.npz` where the reference runs
    the TF Saver, and re-read from there only by `load_model`.
  - dtype: the tuned device path is fp32 (the reference default).  'float64'
    RBMs (Bernoulli, Gaussian, Multinomial) run on the device through
    `bm_rbm64_*` (rbm.py -> RbmEngine64).  A float64 DBM can be
    constructed/initialised but `fit`/`transform` raise.
"""
import json
import os
from copy import deepcopy
from functools import wraps

import numpy as np

from .utils import RNG, write_during_training


def is_param_name(name):
    return not name.startswith('_') and not name.endswith('_')


def is_attribute_name(name):
    return not name.startswith('_') and name.endswith('_')


class BaseMixin(object):
    def __init__(self, *args, **kwargs):
        if args or kwargs:
            raise AttributeError('Invalid parameters: {0}, {1}'.format(args, kwargs))
        super(BaseMixin, self).__init__()


class DtypeMixin(BaseMixin):
    def __init__(self, dtype='float32', *args, **kwargs):
        super(DtypeMixin, self).__init__(*args, **kwargs)
        self.dtype = dtype

    @property
    def _np_dtype(self):
        return getattr(np, self.dtype)


class SeedMixin(BaseMixin):
    def __init__(self, random_seed=None, *args, **kwargs):
        super(SeedMixin, self).__init__(*args, **kwargs)
        self.random_seed = random_seed
        self._rng = RNG(seed=self.random_seed)

    def make_random_seed(self):
        return self._rng.randint(2 ** 31 - 1)


class BaseModel(SeedMixin):
    def get_params(self, deep=True, include_attributes=True):
        params = vars(self)
        p = lambda k: is_param_name(k) or (include_attributes and is_attribute_name(k))
        params = {k: params[k] for k in params if p(k)}
        if deep:
            params = deepcopy(params)
        return params

    def set_params(self, **params):
        for k, v in params.items():
            if (is_param_name(k) or is_attribute_name(k)) and hasattr(self, k):
                setattr(self, k, v)
            else:
                raise ValueError("invalid param name '{0}'".format(k))
        return self

    def _serialize(self, params):
        for k, v in params.items():
            if isinstance(v, np.ndarray):
                if v.size > 1e6:
                    msg = "WARNING: parameter `{0}` won't be serialized because it is too large:"
                    msg += ' ({1:.2f} > 1 Mio elements)'
                    write_during_training(msg.format(k, 1e-6 * v.size))
                    params[k] = None
                else:
                    params[k] = v.tolist()
            elif isinstance(v, (np.floating, np.integer)):
                params[k] = v.item()
        return params

    def _deserialize(self, params):
        return params


def run_on_engine(check_initialized=True, update_seed=False):
    """Stand-in for `run_in_tf_session` (reference base/tf_model.py:10-40).

    update_seed : draw the call's graph seed from the host MT stream, as
        `tf.set_random_seed(model.make_random_seed())` does, and install it as
        the Philox key of every sampling site of this call.
    check_initialized : RuntimeError before `fit`/`init`, as the reference.
    """
    def wrap(f):
        @wraps(f)
        def wrapped_f(model, *args, **kwargs):
            model._graph_seed = model.make_random_seed() if update_seed else None
            if not model.initialized_:
                if check_initialized:
                    raise RuntimeError('`fit` or `init` must be called before calling `{0}`'.format(f.__name__))
                # the reference builds its graph here (`_make_tf_model()`, tf_model.py:31-35) - after the seed above
                model._on_graph_build()
            model._ensure_engine()
            if update_seed:
                model._seed_engine(model._graph_seed)
            try:
                return f(model, *args, **kwargs)
            finally:
                model._join_save()          # checkpoint files of this call are complete when it returns
        return wrapped_f
    return wrap


class EngineModel(BaseModel, DtypeMixin):
    def __init__(self, model_path='tf_model/', paths=None, tf_session_config=None, tf_saver_params=None,
                 json_params=None, *args, **kwargs):
        super(EngineModel, self).__init__(*args, **kwargs)
        # accepted for signature compatibility with the reference (tf_model.py:44-46); unused
        self.tf_saver_params = tf_saver_params or {}
        self._model_dirpath = None
        self._model_filepath = None
        self._params_filepath = None
        self._random_state_filepath = None
        self._train_summary_dirpath = None
        self._val_summary_dirpath = None
        self._tf_meta_graph_filepath = None
        self.update_working_paths(model_path=model_path, paths=paths)
        self.json_params = json_params or {}
        self.json_params.setdefault('sort_keys', True)
        self.json_params.setdefault('indent', 4)
        self.initialized_ = False
        self._engine = None
        self._graph_seed = None
        self._pending_vars = None      # variables read by load_model, uploaded when the engine is built

    # ---- paths (reference tf_model.py:71-99) -------------------------------------
    @staticmethod
    def compute_working_paths(model_path):
        head, tail = os.path.split(model_path)
        if not head:
            head = '.'
        if not head.endswith('/'):
            head += '/'
        if not tail:
            tail = 'model'
        paths = {}
        paths['model_dirpath'] = head
        paths['model_filepath'] = os.path.join(paths['model_dirpath'], tail)
        paths['params_filepath'] = os.path.join(paths['model_dirpath'], 'params.json')
        paths['random_state_filepath'] = os.path.join(paths['model_dirpath'], 'random_state.json')
        paths['train_summary_dirpath'] = os.path.join(paths['model_dirpath'], 'logs/train')
        paths['val_summary_dirpath'] = os.path.join(paths['model_dirpath'], 'logs/val')
        paths['tf_meta_graph_filepath'] = paths['model_filepath'] + '.meta'
        return paths

    def update_working_paths(self, model_path=None, paths=None):
        paths = paths or {}
        if not paths:
            paths = EngineModel.compute_working_paths(model_path=model_path)
        for k, v in paths.items():
            setattr(self, '_{0}'.format(k), v)

    # ---- engine hooks (class specific) -------------------------------------------
    def _make_engine(self):
        """Build the device handle and upload the initial variables (the
        reference's `_make_tf_model` + `global_variables_initializer`)."""
        raise NotImplementedError

    def _seed_engine(self, seed):
        raise NotImplementedError

    def _variables(self):
        """dict name -> ndarray of every checkpointed variable."""
        raise NotImplementedError

    def _ensure_engine(self):
        if self._engine is None:
            if np.dtype(self.dtype) not in (np.dtype(np.float32), np.dtype(np.float64)) and self._needs_device():
                raise NotImplementedError("%s has no device path for dtype='%s' (float32: the tuned path; float64: the "
                                          "compatibility paths bm_rbm64_* / bm_dbm64_*)" % (self.__class__.__name__, self.dtype))
            # Data parallelism is OPT-IN: BM355_DATA_PARALLEL=1 in the environment of a one-process-per-GPU job
            # (RANK / LOCAL_RANK / WORLD_SIZE from the launcher).  Then this process binds to its GPU and joins the
            # job's communicator before the handle is created (boltzmann_machines_amd/parallel.py), and the semantics
            # of fit() change: a global minibatch is world x batch_size rows (N must be a multiple of it), every rank
            # holds n_particles particles (world x n_particles in total), rank 0 alone writes checkpoints and logs,
            # and EVERY RANK MUST MAKE THE SAME PUBLIC CALLS IN THE SAME ORDER (checked at every fit(): a diverging
            # rank raises instead of dead-locking in a collective).  Without the opt-in a launcher's RANK / WORLD_SIZE
            # are ignored: independent per-rank work (hyper-parameter sweeps, pytest under torchrun) stays independent.
            from . import parallel
            self._rank, local_rank, self._world = 0, 0, 1
            if os.environ.get('BM355_DATA_PARALLEL', '0') == '1':
                self._rank, local_rank, self._world = parallel.dist_env()
            self._comm = None
            if self._world > 1:
                from . import _ffi
                try:
                    _ffi.check(_ffi.load().bm_set_device(local_rank))
                    self._comm = parallel.default_comm()
                except _ffi.Bm355Error:
                    if self._needs_device():
                        raise
            self._make_engine()
            if self._pending_vars is not None:
                self._upload_variables(self._pending_vars)
                self._pending_vars = None

    def _needs_device(self):
        return True

    def _on_graph_build(self):
        """what building the TF graph does to HOST state in the reference (nothing, for most models)"""

    def _upload_variables(self, d):
        raise NotImplementedError

    # ---- persistence (reference tf_model.py:117-162) ----------------------------
    def _save_model(self, global_step=None):
        if getattr(self, '_world', 1) > 1 and getattr(self, '_rank', 0) != 0:
            return                      # data-parallel replicas are identical: rank 0 writes the checkpoint
        for dirpath in (self._train_summary_dirpath, self._val_summary_dirpath):
            if not os.path.exists(dirpath):
                os.makedirs(dirpath)
        params = self.get_params(deep=False)
        params = self._serialize(dict(params))
        params['__class_name__'] = self.__class__.__name__
        params_json = json.dumps(params, **self.json_params)
        rng_json = json.dumps(self._rng.get_state()) if self.random_seed is not None else None
        paths = (self._params_filepath, self._random_state_filepath, self._model_filepath + '.npz')
        # The files are written by a background thread while the next epoch trains; every public call joins it before
        # it returns, so callers never see half-written files.  When a snapshot arrives while the previous one is still
        # being written (an epoch shorter than the ~5 ms write of a 784 x 1024 model), it waits in a one-deep slot and
        # REPLACES an older waiting one: the training loop never blocks on the disk, and the newest state is what ends
        # up on it.  Models whose engine can stage a snapshot on the device (`_stage_variables`) do not even stop the
        # stream: the variables are copied device-to-device in stream order into one of two slots and the WRITER reads
        # them back, while the next epoch is already running.
        import threading
        lock = self.__dict__.setdefault('_save_lock', threading.Lock())
        with lock:
            busy = bool(self.__dict__.get('_save_busy'))
            pending = self.__dict__.get('_save_pending')
            # the slot a staged snapshot may use: the waiting job's (it is replaced anyway), else the one the writer
            # is not reading
            if busy and pending is not None and pending[4] is not None:
                slot = pending[4]
            else:
                slot = 1 - self.__dict__.get('_save_inflight_slot', 1) if busy else 0
            staged = self._stage_variables(slot)         # None: no staging, take the snapshot on the host now
            if staged is None:
                variables, slot = self._variables(), None    # (synchronises the stream)
            else:
                variables = staged
            job = (params_json, rng_json, variables, paths, slot)
            if busy:
                self._save_pending = job
                return
            self._save_busy = True
            self._save_inflight_slot = slot if slot is not None else 1
        try:
            self._save_thread = threading.Thread(target=self._save_writer, args=(job, lock), daemon=False)
            self._save_thread.start()
        except (RuntimeError, OSError):     # "can't start new thread": write this checkpoint synchronously
            # (only the failure to START is handled: a KeyboardInterrupt / SystemExit that arrives while the thread is
            # already running must propagate, or two writers would run on the same .tmp files - round-4 advisor)
            self._save_thread = None
            self._save_writer(job, lock)    # (drains a waiting snapshot and clears _save_busy like the thread would)

    def _stage_variables(self, slot):
        """engines with device-side snapshot slots: stage the variables into `slot` and return a callable that reads
        them back (called by the writer thread); None = not supported, the caller snapshots on the host"""
        return None

    @staticmethod
    def _same_checkpoint(a, b):
        if a is None or a[0] != b[0] or a[1] != b[1] or a[3] != b[3]:
            return False
        try:
            va, vb = a[2], b[2]
            if sorted(va.keys()) != sorted(vb.keys()):
                return False
            # ... and the files are still there (deleted or replaced from outside: the reference always rewrites)
            if not all(os.path.isfile(f) for f in (a[3][0], a[3][2])) or (a[1] is not None and not os.path.isfile(a[3][1])):
                return False
            return all(np.array_equal(va[k], vb[k]) for k in va.keys())
        except Exception:       # noqa: BLE001 - anything unusual about the snapshot: write it
            return False

    def _save_writer(self, job, lock):
        while True:
            # every file goes to a temporary name first and is renamed into place: a failed write (disk full,
            # directory removed) never leaves a half-written checkpoint behind, and its exception is kept for
            # _join_save() to re-raise in the calling thread (the reference's synchronous save raises there)
            params_json, rng_json, variables, paths, _slot = job
            try:
                if callable(variables):                  # a staged snapshot: read it back here, off the training thread
                    variables = variables()
                job = (params_json, rng_json, variables, paths, None)
                # fit() saves once more after the last epoch's save (as the reference does): when nothing changed in
                # between, the files on disk already hold exactly this state
                if not self._same_checkpoint(self.__dict__.get('_save_written'), job):
                    tmp = paths[0] + '.tmp'
                    with open(tmp, 'w') as f:
                        f.write(params_json)
                    os.replace(tmp, paths[0])
                    if rng_json is not None:
                        tmp = paths[1] + '.tmp'
                        with open(tmp, 'w') as f:
                            f.write(rng_json)
                        os.replace(tmp, paths[1])
                    # where the reference calls tf.train.Saver.save(session, model_filepath, global_step)
                    tmp = paths[2] + '.tmp.npz'
                    np.savez(tmp, **variables)
                    os.replace(tmp, paths[2])
                    self._save_written = job
            except BaseException as e:      # noqa: BLE001 - handed to the caller by _join_save
                if self.__dict__.get('_save_exc') is None:
                    self._save_exc = e
                self._save_written = None   # nothing is known to be on disk: the next save writes, whatever it holds
            with lock:
                job = self.__dict__.pop('_save_pending', None)
                if job is None:
                    self._save_busy = False
                    return
                self._save_inflight_slot = job[4] if job[4] is not None else 1

    def _join_save(self):
        t = self.__dict__.pop('_save_thread', None)
        if t is not None:
            t.join()                        # (the writer drains the waiting snapshot before it ends)
        e = self.__dict__.pop('_save_exc', None)
        if e is not None:
            raise e

    def _check_lockstep(self, tag):
        """data-parallel mode: every rank must be making the same public call on the same model (a fingerprint of
        class, layer sizes, call counter and `tag` is max- and min-reduced over the ranks); raises on divergence
        instead of letting the collectives of the training loop dead-lock."""
        comm = getattr(self, '_comm', None)
        if comm is None:
            return
        import zlib
        from ._ffi import DeviceArray
        self._dp_calls = getattr(self, '_dp_calls', 0) + 1
        sig = '%s|%s|%d|%s' % (self.__class__.__name__, self._lockstep_signature(), self._dp_calls, tag)
        f = float(zlib.crc32(sig.encode()) & 0xFFFFFF)          # exact in float32
        d = DeviceArray.from_numpy(np.array([f, -f], dtype=np.float32))
        comm.allreduce_max(d, 2)
        hi, lo = d.numpy()
        if hi != f or -lo != f:
            raise RuntimeError('data-parallel ranks diverged at %s (rank %d): every rank must make the same public '
                               'calls on identically configured models' % (sig, self._rank))

    def _lockstep_signature(self):
        return '%s|%s|%s' % (getattr(self, 'n_visible', ''), getattr(self, 'n_hidden', getattr(self, 'n_hiddens', '')),
                             getattr(self, 'batch_size', ''))

    @classmethod
    def load_model(cls, model_path):
        paths = EngineModel.compute_working_paths(model_path)
        with open(paths['params_filepath'], 'r') as params_file:
            params = json.load(params_file)
        class_name = params.pop('__class_name__')
        if class_name != cls.__name__:
            raise RuntimeError("attempt to load {0} with class {1}".format(class_name, cls.__name__))
        model = cls(paths=paths, **{k: params[k] for k in params if is_param_name(k)})
        params = model._deserialize(params)
        model.set_params(**params)
        if os.path.isfile(model._random_state_filepath):
            with open(model._random_state_filepath, 'r') as random_state_file:
                model._rng.set_state(json.load(random_state_file))
        # variables are uploaded once any computation is needed (reference: lazily restored)
        with np.load(model._model_filepath + '.npz') as z:
            model._pending_vars = {k: z[k] for k in z.files}
        return model

    # ---- scalar logs: the TensorBoard-free stand-in for tf.summary (base_rbm.py:520-525,
    # :584-589, dbm.py:636-639): one JSON line per record under logs/train or logs/val ----------
    def _log_scalars(self, kind, step, values):
        values = {k: float(v) for k, v in values.items() if v is not None}
        if not values or (getattr(self, '_world', 1) > 1 and getattr(self, '_rank', 0) != 0):
            return
        d = self._train_summary_dirpath if kind == 'train' else self._val_summary_dirpath
        if not os.path.exists(d):
            os.makedirs(d)
        values['step'] = int(step)
        with open(os.path.join(d, 'scalars.jsonl'), 'a') as f:
            f.write(json.dumps(values, sort_keys=True) + '\n')

    # ---- array dumps: the TensorBoard-free stand-in for tf.summary.image (base_rbm.py:300-306, :429-435,
    # dbm.py:312-322, :531-547): one .npy per quantity and epoch under logs/train, in the layout the reference
    # hands to tf.summary.image ([n, height, width, channels] for filters / particles) ----------------------------
    def _dump_array(self, name, array):
        if getattr(self, '_world', 1) > 1 and getattr(self, '_rank', 0) != 0:
            return
        d = self._train_summary_dirpath
        if not os.path.exists(d):
            os.makedirs(d)
        np.save(os.path.join(d, '%s_epoch%04d.npy' % (name, int(self.epoch_))), np.asarray(array, dtype=np.float32))

    def _as_images(self, rows):
        """[n, V] -> [n, v_shape[0], v_shape[1], v_shape[2]] exactly as the reference reshapes before tf.summary.image"""
        rows = np.asarray(rows)
        return rows.reshape(len(rows), self.v_shape[2], self.v_shape[0], self.v_shape[1]).transpose(0, 2, 3, 1)

    # ---- public API (reference tf_model.py:164-202) ------------------------------
    def _fit(self, X, X_val=None, *args, **kwargs):
        raise NotImplementedError('`fit` is not implemented')

    @run_on_engine(check_initialized=False)
    def init(self):
        if not self.initialized_:
            self.initialized_ = True
            self._save_model()
        return self

    @run_on_engine(check_initialized=False, update_seed=True)
    def fit(self, X, X_val=None, *args, **kwargs):
        self.initialized_ = True
        self._check_lockstep('fit')
        self._fit(X, X_val=X_val, *args, **kwargs)
        self._save_model()
        return self

    @run_on_engine()
    def get_tf_params(self, scope=None):
        """Variables by TF name, as the reference's get_tf_params (tf_model.py:183-202): the variables whose full
        name matches `scope` as a prefix regex (`tf.get_collection(GLOBAL_VARIABLES, scope=scope)`), keyed by that
        name with every occurrence of `scope` and a leading '/' removed."""
        import re
        out = {}
        for _, (tf_name, value) in self._scoped_variables().items():
            if tf_name is None or (scope is not None and not re.match(scope, tf_name)):
                continue
            key = tf_name
            if scope and scope in key:
                key = key.replace(scope, '')
            if key.startswith('/'):
                key = key[1:]
            out[key] = value
        return out

    def _scoped_variables(self):
        """engine variable name -> (full TF variable name in the reference's graph | None, value)"""
        raise NotImplementedError
