"""BernoulliRBM / GaussianRBM with the reference's sklearn-like API on the MI355X engine.

Drop-in for boltzmann_machines/rbm/{base_rbm,rbm}.py of the reference:
same constructor keywords (base_rbm.py:95-105, rbm.py:88-99), `fit`, `init`,
`transform`, `get_tf_params`, `init_from`, `load_model`, `get_params`/
`set_params`, attributes `epoch_`, `iter_`; same schedules (1-based epoch index,
base_rbm.py:535-541), metric cadence (:549-571), validation metrics (:573-590),
free-energy gap (:592-621) and per-epoch checkpoint (:665-666).  The TF graph
of `_make_train_op` (:415-525) is executed by libbm355 (csrc/bm_rbm.hip).

BernoulliRBM, MultinomialRBM (rbm.py:25-65) and GaussianRBM (rbm.py:68-116) in float32 and float64.
"""
import os
import sys

import numpy as np

from . import _ffi
from .base import EngineModel, is_attribute_name, run_on_engine
from .engine import RbmEngine, RbmEngine64
from .utils import epoch_iter, make_list_from, write_during_training
from .utils import philox


def assert_shape(obj, name, desired_shape):
    actual_shape = getattr(obj, name).shape
    if actual_shape != desired_shape:
        raise ValueError('`{0}` has invalid shape {1} != {2}'.format(name, actual_shape, desired_shape))


def assert_len(obj, name, desired_len):
    actual_len = len(getattr(obj, name))
    if actual_len != desired_len:
        raise ValueError('`{0}` has invalid len {1} != {2}'.format(name, actual_len, desired_len))


class BaseRBM(EngineModel):
    """Restricted Boltzmann machine trained with CD-k (reference base_rbm.py:14-94)."""

    _V_UNIT = _ffi.UNIT_BERNOULLI
    _H_UNIT = _ffi.UNIT_BERNOULLI

    def __init__(self,
                 n_visible=784, v_layer_cls=None, v_layer_params=None,
                 n_hidden=256, h_layer_cls=None, h_layer_params=None,
                 W_init=0.01, vb_init=0., hb_init=0., n_gibbs_steps=1,
                 learning_rate=0.01, momentum=0.9, max_epoch=10, batch_size=10, l2=1e-4,
                 sample_v_states=False, sample_h_states=True, dropout=None,
                 sparsity_target=0.1, sparsity_cost=0., sparsity_damping=0.9,
                 dbm_first=False, dbm_last=False,
                 metrics_config=None, verbose=True, save_after_each_epoch=True,
                 display_filters=0, display_hidden_activations=0, v_shape=(28, 28),
                 model_path='rbm_model/', *args, **kwargs):
        super(BaseRBM, self).__init__(model_path=model_path, *args, **kwargs)
        self.n_visible = n_visible
        self.n_hidden = n_hidden

        self.W_init = W_init
        if hasattr(self.W_init, '__iter__'):
            self.W_init = np.asarray(self.W_init)
            assert_shape(self, 'W_init', (self.n_visible, self.n_hidden))
        self.vb_init = vb_init
        if hasattr(self.vb_init, '__iter__'):
            self.vb_init = np.asarray(self.vb_init)
            assert_len(self, 'vb_init', self.n_visible)
        self.hb_init = hb_init
        if hasattr(self.hb_init, '__iter__'):
            self.hb_init = np.asarray(self.hb_init)
            assert_len(self, 'hb_init', self.n_hidden)

        # these can be set by `init_from`
        self._dW_init = None
        self._dvb_init = None
        self._dhb_init = None

        self.n_gibbs_steps = make_list_from(n_gibbs_steps)
        self.learning_rate = make_list_from(learning_rate)
        self.momentum = make_list_from(momentum)
        self.max_epoch = max_epoch
        self.batch_size = batch_size
        self.l2 = l2

        self.sample_h_states = sample_h_states
        self.sample_v_states = sample_v_states
        self.dropout = dropout

        self.sparsity_target = sparsity_target
        self.sparsity_cost = sparsity_cost
        self.sparsity_damping = sparsity_damping

        self.dbm_first = dbm_first
        self.dbm_last = dbm_last

        self.metrics_config = metrics_config or {}
        self.metrics_config.setdefault('l2_loss', False)
        self.metrics_config.setdefault('msre', False)
        self.metrics_config.setdefault('pll', False)
        self.metrics_config.setdefault('feg', False)
        self.metrics_config.setdefault('l2_loss_fmt', '.2e')
        self.metrics_config.setdefault('msre_fmt', '.4f')
        self.metrics_config.setdefault('pll_fmt', '.3f')
        self.metrics_config.setdefault('feg_fmt', '.2f')
        self.metrics_config.setdefault('train_metrics_every_iter', 10)
        self.metrics_config.setdefault('val_metrics_every_epoch', 1)
        self.metrics_config.setdefault('feg_every_epoch', 2)
        self.metrics_config.setdefault('n_batches_for_feg', 10)
        self._train_metrics_names = ('l2_loss', 'msre', 'pll')
        self._val_metrics_names = ('msre', 'pll')

        self.verbose = verbose
        self.save_after_each_epoch = save_after_each_epoch

        assert self.n_hidden >= display_filters
        self.display_filters = display_filters
        assert self.n_hidden >= display_hidden_activations
        self.display_hidden_activations = display_hidden_activations
        self.v_shape = v_shape
        if len(self.v_shape) == 2:
            self.v_shape = (self.v_shape[0], self.v_shape[1], 1)

        # current epoch and iteration
        self.epoch_ = 0
        self.iter_ = 0

    # ---- variables -----------------------------------------------------------------
    def _sigma_vector(self):
        return np.ones(self.n_visible, dtype=self._np_dtype)

    def _initial_variables(self):
        """The initialisers of `_make_vars` (reference base_rbm.py:271-327).

        W ~ tf.random_normal(stddev=W_init, seed=random_seed): with the graph seed of
        the enclosing public call when there is one (`fit`), else TF's
        DEFAULT_GRAPH_SEED (`init()`; pinned by rbm/tests/test_rbm.py:64-67)."""
        dt = self._np_dtype
        V, H = self.n_visible, self.n_hidden
        if hasattr(self.W_init, '__iter__'):
            W = np.asarray(self.W_init, dtype=dt)
        else:
            op_seed = self.random_seed
            if op_seed is None:      # TF: non-deterministic when no seed is given
                op_seed = int(np.random.randint(2 ** 31 - 1))
            graph_seed = self._graph_seed if self._graph_seed is not None else philox.DEFAULT_GRAPH_SEED
            W = (philox.normal(graph_seed, int(op_seed) & 0xFFFFFFFF, (int(op_seed) >> 32) & 0xFFFFFFFF, V * H, dt)
                 * dt(self.W_init)).reshape(V, H).astype(dt)
        vb = np.asarray(self.vb_init, dtype=dt) if hasattr(self.vb_init, '__iter__') \
            else np.repeat(dt(self.vb_init), V)
        hb = np.asarray(self.hb_init, dtype=dt) if hasattr(self.hb_init, '__iter__') \
            else np.repeat(dt(self.hb_init), H)
        z = lambda a, shape: np.zeros(shape, dtype=dt) if a is None else np.asarray(a, dtype=dt).reshape(shape)
        return dict(W=W, vb=vb, hb=hb, dW=z(self._dW_init, (V, H)), dvb=z(self._dvb_init, (V,)),
                    dhb=z(self._dhb_init, (H,)), q_means=np.zeros(H, dtype=dt), sigma=self._sigma_vector())

    _VAR_SCOPES = (('W', 'weights'), ('vb', 'weights'), ('hb', 'weights'),
                   ('dW', 'grads_accumulators'), ('dvb', 'grads_accumulators'), ('dhb', 'grads_accumulators'),
                   ('q_means', 'hidden_activations_means'), ('sigma', 'input_data'))

    def _make_engine(self):
        variables = self._initial_variables()
        f64 = np.dtype(self.dtype) == np.float64
        if np.dtype(self.dtype) == np.float32 or f64:
            self._engine = (RbmEngine64 if f64 else RbmEngine)(
                                     self.n_visible, self.n_hidden, v_unit=self._V_UNIT,
                                     sample_v_states=self.sample_v_states, sample_h_states=self.sample_h_states,
                                     dbm_first=self.dbm_first, dbm_last=self.dbm_last, max_batch=self.batch_size,
                                     l2=self.l2, sparsity_target=self.sparsity_target,
                                     sparsity_cost=self.sparsity_cost, sparsity_damping=self.sparsity_damping,
                                     dropout=self.dropout, h_unit=self._H_UNIT,
                                     n_samples=getattr(self, 'n_samples', 0))
            self._upload_variables(variables)
            # multi-GPU job (one process per GPU): data-parallel CD-k, rank r takes rows [r*batch_size, ...) of every
            # global minibatch of world*batch_size rows; ONE all-reduce(sum) of the fused [dW|dvb|dhb|q] buffer per
            # update through the library's communicator (parallel.DataParallelRBM)
            self._dp = None
            if getattr(self, '_comm', None) is not None and not f64:
                from . import parallel
                self._dp = parallel.DataParallelRBM(self._engine, self._rank, self._world, self.batch_size,
                                                    parallel.native_allreduce_on_engine_stream(self._engine, self._comm))
        else:
            raise NotImplementedError("%s: dtype must be 'float32' or 'float64' (got %r)" % (self.__class__.__name__, self.dtype))

    def _needs_device(self):
        return False

    def _on_device(self):
        if not isinstance(self._engine, (RbmEngine, RbmEngine64)):
            raise NotImplementedError("no device path for %s with dtype='%s'" % (self.__class__.__name__, self.dtype))
        return self._engine

    def _to_device(self, X, slot=None):
        """host array -> DeviceArray in the engine's dtype; `slot` names a buffer of this model that is reused
        across calls (the training / validation set of repeated fit() calls)"""
        dt = self._engine.dtype
        if slot is None:
            return _ffi.DeviceArray.from_numpy(np.ascontiguousarray(X, dtype=dt), dt)
        pool = self.__dict__.setdefault('_dev_pool', {})
        pool[slot] = _ffi.DeviceArray.from_numpy_reusing(pool.get(slot), X, dt)
        return pool[slot]

    def _upload_variables(self, d):
        for name, _ in self._VAR_SCOPES:
            if name in d:
                self._engine.set(name, d[name])

    def _seed_engine(self, seed):
        if isinstance(self._engine, (RbmEngine, RbmEngine64)):
            self._engine.seed(seed)

    def _variables(self):
        return {name: self._engine.get(name) for name, _ in self._VAR_SCOPES}

    def _scoped_variables(self):
        # the reference's variable names (base_rbm.py:271-327); `sigma` is a variable of the GaussianRBM only, created
        # under a second 'input_data' name scope (rbm.py:101-105 -> 'input_data_1/sigma')
        names = {name: '%s/%s' % (scope, name) for name, scope in self._VAR_SCOPES}
        names['sigma'] = 'input_data_1/sigma' if self._V_UNIT == _ffi.UNIT_GAUSSIAN else None
        return {name: (names[name], self._engine.get(name)) for name, _ in self._VAR_SCOPES}

    def _stage_variables(self, slot):
        """checkpoint snapshot without stopping the stream (bm_rbm_stage): float32 engine only; BM355_STAGED_SAVE=0
        restores the host-side snapshot"""
        eng = self._engine
        if not isinstance(eng, RbmEngine) or os.environ.get('BM355_STAGED_SAVE', '1') == '0':
            return None
        eng.stage(slot)
        names = [name for name, _ in self._VAR_SCOPES]
        return lambda: {name: eng.get_staged(slot, name) for name in names}

    def set_params(self, **params):
        # the handle bakes in the graph constants: rebuild it if one of them changes
        rebuild = {'batch_size', 'l2', 'dropout', 'sample_v_states', 'sample_h_states', 'sparsity_target',
                   'sparsity_cost', 'sparsity_damping', 'dbm_first', 'dbm_last'}
        super(BaseRBM, self).set_params(**params)
        if self._engine is not None and isinstance(self._engine, (RbmEngine, RbmEngine64)) and rebuild & set(params):
            self._pending_vars = self._variables()
            self._engine.close()
            self._engine = None
        return self

    # ---- schedules (reference base_rbm.py:533-547) -----------------------------------
    def _schedule(self, values):
        return values[min(self.epoch_, len(values) - 1)]

    def _feed(self):
        return (float(self._schedule(self.learning_rate)), float(self._schedule(self.momentum)),
                int(self._schedule(self.n_gibbs_steps)))

    # ---- training loop (reference base_rbm.py:549-666) -------------------------------
    def _train_epoch(self, Xd, N, after_first=None, defer=False):
        """one epoch of updates.  `after_first`: called once, right after the first engine call of the epoch (or at its start
        when the epoch opens with a metrics iteration) - `_fit` hands in the PREVIOUS epoch's report there, so that its one
        wait for the device and its host work run under this epoch's first run of updates instead of in front of it.
        `defer=True`: return a callable that collects the epoch's train metrics later instead of the metrics themselves."""
        eng = self._on_device()
        names = sorted(m for m in self._train_metrics_names if self.metrics_config[m])
        results = {m: [] for m in names}
        lr, mom, k = self._feed()
        every = self.metrics_config['train_metrics_every_iter']
        if getattr(self, '_dp', None) is not None:
            # data-parallel epoch: global minibatches of world * batch_size rows (train metrics are not fetched:
            # they would be rank-local numbers; validation metrics are evaluated on this rank's replica)
            step_rows = self.batch_size * self._world
            if N % step_rows:
                raise ValueError('data-parallel fit: {0} rows are not a multiple of batch_size x world_size = {1}'
                                 .format(N, step_rows))
            for start in range(0, N, step_rows):
                self.iter_ += 1
                self._dp.train_step(Xd, lr, mom, k, row=start + self._rank * self.batch_size)
            if after_first is not None:
                after_first()
            none = {m: None for m in names}
            return (lambda: none) if defer else none
        # runs of batches without a metrics fetch go to the engine as ONE call (bm_rbm_train_epoch loops in
        # C: same launches, same RNG call counters, no Python per batch)
        # ... and the metrics iterations leave their sums in a pinned ring (train_step_metrics_async): the reference
        # only uses the epoch mean (base_rbm.py:571), so the host waits for the stream once per epoch
        run_start, fused = None, hasattr(eng, 'train_epoch')
        deferred, pending = hasattr(eng, 'train_step_metrics_async'), 0

        def collect():
            for out in eng.collect_metrics():
                vals = dict(msre=out[0], pll=out[1], l2_loss=out[2])
                for m in names:
                    results[m].append(vals[m])
            return 0

        def first_done():
            nonlocal after_first
            if after_first is not None:
                f, after_first = after_first, None
                f()
        deferred_result = None
        try:
            for start in range(0, N, self.batch_size):
                B = min(self.batch_size, N - start)
                self.iter_ += 1
                if self.iter_ % every == 0:
                    if run_start is not None:
                        eng.train_epoch(Xd, start - run_start, self.batch_size, lr, mom, k, row=run_start)
                        run_start = None
                    first_done()          # (before this epoch's first fetch: the ring then holds the previous epoch's only)
                    if deferred:
                        if pending >= eng.MAX_PENDING_METRICS:
                            pending = collect()
                        eng.train_step_metrics_async(Xd, B, lr, mom, k, row=start)
                        pending += 1
                    else:
                        out = eng.train_step_metrics(Xd, B, lr, mom, k, row=start)
                        vals = dict(msre=out[0], pll=out[1], l2_loss=out[2])
                        for m in names:
                            results[m].append(vals[m])
                elif fused:
                    if run_start is None:
                        run_start = start
                else:
                    eng.train_step(Xd, B, lr, mom, k, row=start)
                    first_done()
            if run_start is not None:
                eng.train_epoch(Xd, N - run_start, self.batch_size, lr, mom, k, row=run_start)
            first_done()
            if defer and deferred:
                n_pending, pending = pending, 0            # (the `finally` below must not drain what the caller will collect)

                def deferred_result():
                    if n_pending:
                        collect()
                    return {m: (np.mean(r) if r else None) for m, r in results.items()}
            elif pending:
                pending = collect()
        finally:
            # An epoch that aborts BEFORE its first run of updates was queued still owes the previous epoch's report (its
            # fetches are the only ones in the ring then): make it now - progress line, scalar logs - instead of dropping
            # it with the drain below (round-5 advisor).  A report that cannot be made any more is announced, never silent.
            if after_first is not None and sys.exc_info()[0] is not None:
                try:
                    first_done()
                except Exception as e:       # noqa: BLE001
                    sys.stderr.write('fit: the report of the previous epoch was lost with the aborted one (%s)\n' % (str(e)[:200],))
            # an aborted epoch (KeyboardInterrupt, an engine error in a later batch) must not leave its deferred fetches
            # in the ring: the next epoch's collect() would average them into ITS metrics (round-4 advisor)
            if deferred and pending:
                try:
                    eng.collect_metrics()
                except Exception:
                    pass
        if deferred_result is not None:
            return deferred_result
        out = {m: (np.mean(r) if r else None) for m, r in results.items()}
        return (lambda: out) if defer else out

    def _run_val_metrics(self, Xvd, N):
        eng = self._on_device()
        names = sorted(m for m in self._val_metrics_names if self.metrics_config[m])
        results = {m: [] for m in names}
        _, _, k = self._feed()
        for start in range(0, N, self.batch_size):
            B = min(self.batch_size, N - start)
            out = eng.metrics(Xvd, B, k, row=start)
            vals = dict(msre=out[0], pll=out[1])
            for m in names:
                results[m].append(vals[m])
        return {m: (np.mean(r) if r else None) for m, r in results.items()}

    def _run_feg(self, Xd, N, Xvd, Nv):
        """Free-energy gap between validation and training subsets (reference base_rbm.py:592-621)."""
        eng = self._on_device()
        nb = self.metrics_config['n_batches_for_feg']
        train_fes, val_fes = [], []
        for b, start in zip(range(nb), range(0, N, self.batch_size)):
            train_fes.append(eng.free_energy(Xd, min(self.batch_size, N - start), row=start))
        for b, start in zip(range(nb), range(0, Nv, self.batch_size)):
            val_fes.append(eng.free_energy(Xvd, min(self.batch_size, Nv - start), row=start))
        return np.mean(val_fes) - np.mean(train_fes)

    def _fit(self, X, X_val=None, *args, **kwargs):
        self._on_device()
        Xd = self._to_device(X, 'fit_X')
        N = len(X)
        Xvd, Nv = None, 0
        if X_val is not None:
            Xvd, Nv = self._to_device(X_val, 'fit_X_val'), len(X_val)
        # The epoch's train metrics are device sums in a pinned ring (train_step_metrics_async); reading them is the ONE host wait
        # of an epoch.  Where nothing else needs the device idle at the epoch's end (no validation fetch, no display dump), the
        # checkpoint snapshot is staged in stream order at once and the REPORT of the epoch - wait, scalar logs, progress line -
        # is made after the next epoch's first run of updates has been queued (`after_first`): the device never waits for the
        # host's epoch-end work (75 -> 69 us per update with the reference's default cadence of one fetch per 10 updates).
        try:
            self._fit_epochs(X, X_val, Xd, N, Xvd, Nv)
        except BaseException:
            # fetches of an epoch whose report was still to come must not reach a later call's metrics (round-4 advisor)
            try:
                if hasattr(self._engine, 'collect_metrics'):
                    self._engine.collect_metrics()
            except Exception:
                pass
            raise
        self._engine.sync()

    def _fit_epochs(self, X, X_val, Xd, N, Xvd, Nv):
        report_later = None
        for self.epoch_ in epoch_iter(start_epoch=self.epoch_, max_epoch=self.max_epoch, verbose=self.verbose):
            val_now = X_val is not None and self.epoch_ % self.metrics_config['val_metrics_every_epoch'] == 0
            feg_now = X_val is not None and self.metrics_config['feg'] and self.epoch_ % self.metrics_config['feg_every_epoch'] == 0
            pipelined = (not val_now and not feg_now and not self.display_filters and not self.display_hidden_activations
                         and getattr(self, '_dp', None) is None)
            train_later = self._train_epoch(Xd, N, after_first=report_later, defer=True)
            report_later = None

            def report(train_later=train_later, epoch=self.epoch_, it=self.iter_, val_now=val_now, feg_now=feg_now):
                train_results = train_later()
                val_results = self._run_val_metrics(Xvd, Nv) if val_now else {}
                feg = self._run_feg(Xd, N, Xvd, Nv) if feg_now else None
                self._log_scalars('train', it, dict(train_results, epoch=epoch))
                self._log_scalars('val', it, dict(val_results, feg=feg))
                if self.verbose:
                    self._report_epoch(train_results, val_results, feg, epoch=epoch)
            if pipelined:
                if self.save_after_each_epoch:
                    self._save_model(global_step=self.epoch_)       # (the snapshot is taken in stream order, here)
                report_later = report
            else:
                report()
                self._display_dumps(X)
                if self.save_after_each_epoch:
                    self._save_model(global_step=self.epoch_)
        if report_later is not None:
            report_later()

    def _display_dumps(self, X):
        """`display_filters` / `display_hidden_activations` (base_rbm.py:300-306, :429-435): where the reference adds
        image summaries to the train summary, one .npy per epoch goes to logs/train: `W_filters` [n, h, w, c] (the
        first n columns of W as images, the reference's transposes) and `hidden_activation_means` [batch, n] (means
        of the first n hidden units for the first minibatch, computed on the host from the fetched parameters so that
        the device RNG stream of the run does not depend on the display options; the reference shows the last Gibbs
        step's means instead)."""
        if not (self.display_filters or self.display_hidden_activations):
            return
        W = np.asarray(self._engine.get('W'), dtype=np.float64)
        if self.display_filters and self.n_visible == int(np.prod(self.v_shape)):
            self._dump_array('W_filters', self._as_images(W.T[:self.display_filters]))
        if self.display_hidden_activations:
            n = self.display_hidden_activations
            Xb = np.asarray(X[:self.batch_size], dtype=np.float64)
            if self._V_UNIT == _ffi.UNIT_GAUSSIAN:
                Xb = Xb / np.asarray(self._sigma_vector(), dtype=np.float64)
            z = (1.0 + float(bool(self.dbm_first))) * (Xb.dot(W[:, :n]) + np.asarray(self._engine.get('hb'), dtype=np.float64)[:n])
            if self._H_UNIT == _ffi.UNIT_MULTINOMIAL:      # means of the whole layer are needed for the softmax
                zz = (1.0 + float(bool(self.dbm_first))) * (Xb.dot(W) + np.asarray(self._engine.get('hb'), dtype=np.float64))
                e = np.exp(zz - zz.max(axis=1, keepdims=True))
                hm = (self.n_samples if hasattr(self, 'n_samples') else 1) * (e / e.sum(axis=1, keepdims=True))[:, :n]
            else:
                hm = 1.0 / (1.0 + np.exp(-z))
            self._dump_array('hidden_activation_means', hm)

    def _report_epoch(self, train_results, val_results, feg, epoch=None):
        """one progress line per epoch in the reference's wording (`epoch: 3/10; msre: ...; val.pll: ...; feg: ...`,
        base_rbm.py:652-666), formats from metrics_config['<metric>_fmt']"""
        fmt = lambda name, value: format(value, self.metrics_config[name + '_fmt'])
        parts = ['epoch: %*d/%d' % (len(str(self.max_epoch)), self.epoch_ if epoch is None else epoch, self.max_epoch)]
        parts += ['%s: %s' % (m, fmt(m, v)) for m, v in sorted(train_results.items()) if v is not None]
        parts += ['val.%s: %s' % (m, fmt(m, v)) for m, v in sorted(val_results.items()) if v is not None]
        line = '; '.join(parts)
        if feg is not None:
            line += ' ; feg: ' + fmt('feg', feg)
        write_during_training(line)

    def init_from(self, rbm):
        """Warm start from another RBM of the same class: its weights become this model's initialisers, its
        momentum buffers the initial accumulators, and its run-time attributes (epoch_, iter_, ...) carry over
        (reference base_rbm.py:668-685)."""
        if type(self) != type(rbm):
            raise ValueError('an attempt to initialize `{0}` from `{1}`'
                             .format(type(self).__name__, type(rbm).__name__))
        w, acc = rbm.get_tf_params(scope='weights'), rbm.get_tf_params(scope='grads_accumulators')
        self.W_init, self.vb_init, self.hb_init = w['W'], w['vb'], w['hb']
        self._dW_init, self._dvb_init, self._dhb_init = acc['dW'], acc['dvb'], acc['dhb']
        for name, value in vars(rbm).items():
            if is_attribute_name(name):
                setattr(self, name, value)

    @run_on_engine(update_seed=True)
    def transform(self, X, np_dtype=None):
        """Hidden activation probabilities at the END of the k-step chain, stochastic
        (reference base_rbm.py:687-700 with the op of :438-440)."""
        np_dtype = np_dtype or self._np_dtype
        eng = self._on_device()
        N = len(X)
        Xd = self._to_device(X)
        Hd = _ffi.DeviceArray((N, self.n_hidden), eng.dtype)
        _, _, k = self._feed()
        for start in range(0, N, self.batch_size):
            eng.transform(Xd, min(self.batch_size, N - start), k, Hd, row=start, out_row=start)
        eng.sync()
        return Hd.numpy().astype(np_dtype)


class BernoulliRBM(BaseRBM):
    """RBM with Bernoulli visible and hidden units (reference rbm/rbm.py:10-22)."""

    def __init__(self, model_path='b_rbm_model/', *args, **kwargs):
        super(BernoulliRBM, self).__init__(model_path=model_path, *args, **kwargs)


class MultinomialRBM(BaseRBM):
    """RBM with Bernoulli visible and a single Multinomial hidden unit (= `n_samples` softmax units
    with tied weights; reference rbm/rbm.py:25-65, layers.py:54-70).

    Parameters
    ----------
    n_hidden : int
        Number of possible states of the multinomial unit.
    n_samples : int
        Number of softmax units with shared weights (= draws from one softmax unit).

    Hidden means are `n_samples * softmax(vW + hb)`, hidden states the counts of `n_samples`
    categorical draws; `transform` returns the means divided by `n_samples` (rbm.py:62-65).
    """

    _H_UNIT = _ffi.UNIT_MULTINOMIAL

    def __init__(self, n_samples=100, model_path='m_rbm_model/', *args, **kwargs):
        self.n_samples = n_samples
        super(MultinomialRBM, self).__init__(model_path=model_path, *args, **kwargs)

    def transform(self, *args, **kwargs):
        H = super(MultinomialRBM, self).transform(*args, **kwargs)
        H /= float(self.n_samples)
        return H


class GaussianRBM(BaseRBM):
    """RBM with Gaussian visible (fixed `sigma`) and Bernoulli hidden units
    (reference rbm/rbm.py:68-116)."""

    _V_UNIT = _ffi.UNIT_GAUSSIAN

    def __init__(self, learning_rate=1e-3, sigma=1., model_path='g_rbm_model/', *args, **kwargs):
        self.sigma = sigma
        super(GaussianRBM, self).__init__(learning_rate=learning_rate, model_path=model_path, *args, **kwargs)
        if hasattr(self.sigma, '__iter__'):
            self._sigma_tmp = self.sigma = np.asarray(self.sigma)
        else:
            self._sigma_tmp = np.repeat(self.sigma, self.n_visible)

    def _sigma_vector(self):
        return np.asarray(self._sigma_tmp, dtype=self._np_dtype)


def logit_mean(X):
    """log(p / (1 - p)) of the per-feature mean, clipped (reference rbm/rbm.py:119-123)."""
    p = np.mean(X, axis=0)
    p = np.clip(p, 1e-7, 1. - 1e-7)
    q = np.log(p / (1. - p))
    return q
