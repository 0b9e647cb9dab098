"""Deep Boltzmann Machine with the reference's API on the MI355X engine.

Drop-in for boltzmann_machines/dbm.py of the reference: constructor keywords
(:89-99), `load_rbms` weight composition (:207-231, :266-291), `fit`, `transform`
(:859-872), `reconstruct` (:874-885), `sample_v` (:887-897), `log_Z` (:899-939),
`log_proba` (:941-957), `get_tf_params`, `load_model`, attributes `epoch_`, `iter_`,
`n_layers_`, `n_visible_`, `n_hiddens_`, `n_samples_generated_`.  The TF graph
(mean-field, PCD, train op, AIS, ELBO) is executed by libbm355 (csrc/bm_dbm.hip).

Unit types: Bernoulli or Gaussian visible layer; Bernoulli or Multinomial hidden layers (the CIFAR
model of examples/dbm_cifar.py is Gaussian-Bernoulli-Multinomial; `log_Z` / `log_proba` need all-Bernoulli
layers like the reference, dbm.py:925-927, :947-948).
"""
import os

import numpy as np

from . import _ffi
from .base import EngineModel, run_on_engine
from .engine import DbmEngine, DbmEngine64
from .rbm import GaussianRBM
from .utils import (epoch_iter, make_list_from, write_during_training,
                    log_sum_exp, log_mean_exp, log_diff_exp, log_std_exp)
from .utils import philox

# Philox sites of the host-evaluated initialisers (`layer.init`, dbm.py:362-383)
_SITE_V_INIT, _SITE_V_NEW_INIT, _SITE_H_INIT = 20, 21, 22


def as_device(X, dtype=np.float32):
    """host ndarray -> DeviceArray in the engine's dtype; a DeviceArray passes through"""
    return X if isinstance(X, _ffi.DeviceArray) else _ffi.DeviceArray.from_numpy(np.asarray(X), dtype)


class DBM(EngineModel):
    def __init__(self, rbms=None,
                 n_particles=100, v_particle_init=None, h_particles_init=None,
                 n_gibbs_steps=5, max_mf_updates=10, mf_tol=1e-7,
                 learning_rate=0.0005, momentum=0.9, max_epoch=10, batch_size=100,
                 l2=0., max_norm=np.inf,
                 sample_v_states=True, sample_h_states=None,
                 sparsity_target=0.1, sparsity_cost=0., sparsity_damping=0.9,
                 train_metrics_every_iter=10, val_metrics_every_epoch=1,
                 verbose=False, save_after_each_epoch=True,
                 display_filters=0, display_particles=0, v_shape=(28, 28),
                 model_path='dbm_model/', *args, **kwargs):
        super(DBM, self).__init__(model_path=model_path, *args, **kwargs)
        self.n_layers_ = len(rbms) if rbms is not None else None
        self.n_visible_ = None
        self.n_hiddens_ = []
        self._rbms = None
        self._W_init = self._vb_init = self._hb_init = None
        self.v_unit_ = _ffi.UNIT_BERNOULLI     # visible unit type (attribute: restored by load_model)
        self.h_units_ = []                     # hidden layer kinds / multinomial draw counts (layers.py:39-70)
        self.h_n_samples_ = []
        self._sigma_init = None
        self.load_rbms(rbms)

        self.n_particles = n_particles
        self._v_particle_init = v_particle_init
        self._h_particles_init = h_particles_init

        self.n_gibbs_steps = make_list_from(n_gibbs_steps)
        self.max_mf_updates = max_mf_updates
        self.mf_tol = mf_tol

        self.learning_rate = make_list_from(learning_rate)
        self.momentum = make_list_from(momentum)
        self.max_epoch = max_epoch
        self.batch_size = batch_size
        self.l2 = l2
        self.max_norm = max_norm

        self.sample_v_states = sample_v_states
        self.sample_h_states = sample_h_states or [True] * (self.n_layers_ or 0)

        self.sparsity_target = make_list_from(sparsity_target)
        self.sparsity_cost = make_list_from(sparsity_cost)
        if self.n_layers_ is not None and self.n_layers_ > 1:
            for x in (self.sparsity_target, self.sparsity_cost):
                if len(x) == 1:
                    x *= self.n_layers_
        self.sparsity_damping = sparsity_damping

        self.train_metrics_every_iter = train_metrics_every_iter
        self.val_metrics_every_epoch = val_metrics_every_epoch
        self.verbose = verbose
        self.save_after_each_epoch = save_after_each_epoch

        for nh in self.n_hiddens_:
            assert nh >= display_filters
        self.display_filters = display_filters
        assert display_particles <= self.n_particles
        self.display_particles = display_particles
        self.v_shape = v_shape
        if len(self.v_shape) == 2:
            self.v_shape = (self.v_shape[0], self.v_shape[1], 1)

        self.epoch_ = 0
        self.iter_ = 0
        self.n_samples_generated_ = 0
        # log_Z accumulation (not a constructor keyword of the reference: set_ais_accumulation / BM355_AIS_LITERAL=1)
        self._ais_literal = os.environ.get('BM355_AIS_LITERAL', '0') == '1'
        # sigmoid of the Bernoulli layers (set_mean_field_arithmetic / BM355_SIGMOID_LITERAL=1)
        self._sigmoid_literal = os.environ.get('BM355_SIGMOID_LITERAL', '0') == '1'

    # ---- composition from pre-trained RBMs (reference dbm.py:207-231) ------------------
    def load_rbms(self, rbms):
        if rbms is not None:
            self._rbms = rbms
            # `self._h_layers = [rbm._h_layer for rbm in self._rbms]` (dbm.py:226): the hidden layer of every RBM
            self.h_units_ = [int(getattr(rbm, '_H_UNIT', _ffi.UNIT_BERNOULLI)) for rbm in rbms]
            self.h_n_samples_ = [int(getattr(rbm, 'n_samples', 0) or 0) if u == _ffi.UNIT_MULTINOMIAL else 0
                                 for rbm, u in zip(rbms, self.h_units_)]
            self.n_layers_ = len(self._rbms)
            self.n_visible_ = self._rbms[0].n_visible
            self.n_hiddens_ = [rbm.n_hidden for rbm in self._rbms]
            self._W_init, self._vb_init, self._hb_init = [], [], []
            for i in range(self.n_layers_):
                weights = self._rbms[i].get_tf_params(scope='weights')
                self._W_init.append(weights['W'])
                self._vb_init.append(weights['vb'])
                self._hb_init.append(weights['hb'])
            if isinstance(self._rbms[0], GaussianRBM):
                self.v_unit_ = _ffi.UNIT_GAUSSIAN
                self._sigma_init = self._rbms[0]._sigma_vector()

    def _composed_weights(self):
        """`_make_vars` (reference dbm.py:266-291): halve the intermediate RBMs' weights and
        average the biases that two RBMs give to the same layer."""
        L = self.n_layers_
        W_init, hb_init = [], []
        dt = self._np_dtype
        vb_init = np.array(self._vb_init[0], dtype=dt)
        for i in range(L):
            W = np.array(self._W_init[i], dtype=dt)
            vb = np.array(self._vb_init[i], dtype=dt)
            hb = np.array(self._hb_init[i], dtype=dt)
            if 0 < i < L - 1:
                W *= 0.5
                vb *= 0.5
                hb *= 0.5
            W_init.append(W)
            if i == 0:
                hb_init.append(0.5 * hb)
            else:
                hb_init[i - 1] += 0.5 * vb
                hb_init.append(0.5 * hb if i < L - 1 else hb)
        return W_init, vb_init, hb_init

    @staticmethod
    def _sfx(i):
        return '' if i == 0 else '_%d' % i

    def _var_names(self):
        L = self.n_layers_
        names = [('vb', 'weights'), ('dvb', 'grads_accumulators'), ('sigma', 'input_data'),
                 ('v', 'negative_particles'), ('v_new', 'negative_particles')]
        for i in range(L):
            s = self._sfx(i)
            names += [('W' + s, 'weights'), ('hb' + s, 'weights'),
                      ('dW' + s, 'grads_accumulators'), ('dhb' + s, 'grads_accumulators'),
                      ('mu' + s, 'variational_params'), ('mu_new' + s, 'variational_params'),
                      ('q_means' + s, 'hidden_means_accumulators'), ('mu_means' + s, 'hidden_means_accumulators'),
                      ('h' + s, 'negative_particles'), ('h_new' + s, 'negative_particles')]
        return names

    def _initial_variables(self):
        W_init, vb_init, hb_init = self._composed_weights()
        L, M, V = self.n_layers_, self.n_particles, self.n_visible_
        seed = self._graph_seed if self._graph_seed is not None else philox.DEFAULT_GRAPH_SEED
        d = dict(vb=vb_init)
        dt = self._np_dtype
        if self._sigma_init is not None:
            d['sigma'] = np.asarray(self._sigma_init, dtype=dt)

        def v_init(site):       # `layer.init`: U[0,1) reals for Bernoulli (layers.py:43-45), N(0,1)*sigma for Gaussian
            if self.v_unit_ == _ffi.UNIT_GAUSSIAN:
                return philox.normal(seed, site, 0, M * V, dtype=dt).reshape(M, V) * d.get('sigma', dt(1.))
            return philox.uniform(seed, site, 0, M * V, dtype=dt).reshape(M, V)
        d['v'] = np.asarray(self._v_particle_init, dtype=dt) if self._v_particle_init is not None \
            else v_init(_SITE_V_INIT)
        d['v_new'] = v_init(_SITE_V_NEW_INIT)
        for i in range(L):
            s, n = self._sfx(i), self.n_hiddens_[i]
            d['W' + s], d['hb' + s] = W_init[i], hb_init[i]
            def h_init(site):       # BernoulliLayer.init: U[0,1) (layers.py:43-45); MultinomialLayer.init: the same
                t = philox.uniform(seed, site, 0, M * n, dtype=dt).reshape(M, n)      # divided by the sum over the WHOLE tensor
                if self._h_unit(i) == _ffi.UNIT_MULTINOMIAL:                # (layers.py:59-63)
                    t = (t / np.sum(t, dtype=dt)).astype(dt)
                return t
            if self._h_particles_init is not None:
                d['h' + s] = np.asarray(self._h_particles_init[i], dtype=dt).reshape(M, n)
            else:
                d['h' + s] = h_init(_SITE_H_INIT + 2 * i)
            d['h_new' + s] = h_init(_SITE_H_INIT + 2 * i + 1)
        return d

    def _h_unit(self, i):
        return self.h_units_[i] if self.h_units_ else _ffi.UNIT_BERNOULLI

    def _all_bernoulli(self):
        return self.v_unit_ == _ffi.UNIT_BERNOULLI and all(self._h_unit(i) == _ffi.UNIT_BERNOULLI
                                                           for i in range(self.n_layers_))

    # ---- engine hooks ---------------------------------------------------------------------
    def _make_engine(self):
        if self.n_layers_ is None:
            raise RuntimeError('DBM has no layers: pass `rbms` or use `load_model`')
        f64 = np.dtype(self.dtype) == np.float64
        if np.dtype(self.dtype) != np.float32 and not f64:
            raise NotImplementedError("DBM: dtype must be 'float32' or 'float64' (got %r)" % (self.dtype,))
        if f64:
            # DBM(dtype='float64') (base/mixin.py:14-25): the float64 compatibility path (csrc/bm_dbm64.hip)
            if any(self._h_unit(i) != _ffi.UNIT_BERNOULLI for i in range(self.n_layers_)):
                raise NotImplementedError('float64 DBM: Bernoulli hidden layers only (Multinomial layers run in float32)')
            if getattr(self, '_comm', None) is not None:
                raise NotImplementedError('float64 DBM: single-process only')
        self._engine = (DbmEngine64 if f64 else DbmEngine)(self.n_visible_, self.n_hiddens_, v_unit=self.v_unit_,
                                 sample_v_states=self.sample_v_states, sample_h_states=self.sample_h_states,
                                 n_particles=self.n_particles, batch_size=self.batch_size,
                                 max_mf_updates=self.max_mf_updates, mf_tol=self.mf_tol, l2=self.l2,
                                 max_norm=self.max_norm, sparsity_target=self.sparsity_target,
                                 sparsity_cost=self.sparsity_cost, sparsity_damping=self.sparsity_damping,
                                 h_units=self.h_units_ or None, n_samples=self.h_n_samples_ or None)
        # opt-in speed mode for log_Z (AIS): exact-product bf16 x 3 for the {0,1}-state contractions; tolerance parity
        # (DESIGN.md 3.9).  The reference API has no switch for it, so it is read from the environment.
        if os.environ.get('BM355_FAST_BINARY', '0') == '1' and not self._sigmoid_literal and not f64:
            self._engine.set_fast_binary(True)
        if self._sigmoid_literal and not f64:
            self._engine.set_sigmoid_literal(True)
        if self._pending_vars is None:          # fresh model (load_model uploads its checkpoint instead)
            self._upload_variables(self._initial_variables())
        # Multi-GPU job (one process per GPU): rank r owns rows [r*batch_size, ...) of every global
        # minibatch of world*batch_size rows and particles [r*n_particles, ...) of world*n_particles; one
        # all-reduce(sum) of the fused gradient buffer per update, the mean-field residual all-reduced (max) on
        # the device per sweep, replicas bit-identical.  Every rank must make the same public calls.
        self._dp = None
        if getattr(self, '_comm', None) is not None:
            from . import parallel
            self._dp = parallel.DataParallelDBM(self._engine, self._rank, self._world,
                                                parallel.native_allreduce_on_engine_stream(self._engine, self._comm),
                                                comm=self._comm)

    def _on_graph_build(self):
        # `_make_ais` draws the op-level seed of x_0 from the host MT stream WHILE THE GRAPH IS BUILT
        # (`Bernoulli(logits).sample(seed=self.make_random_seed())`, dbm.py:701): the first fit() / init() of a DBM
        # advances the stream by one more draw than the call's own graph seed, and every later public call sees the
        # seed sequence shifted by it.  (The engine addresses x_0 by site 13 of the log_Z call's key, DESIGN.md 4;
        # the drawn value itself is not used.)  Found by running the reference's graph builders, round 4.
        self._ais_op_seed = self.make_random_seed()

    def _upload_variables(self, d):
        for name, _ in self._var_names():
            if name in d:
                self._engine.set(name, d[name])

    def _seed_engine(self, seed):
        self._engine.seed(seed)

    def _variables(self):
        return {name: self._engine.get(name) for name, _ in self._var_names()}

    def _scoped_variables(self):
        """the reference's variable names (dbm.py:294-383): layer i > 0 gets TF's `_i` suffix, the hidden particles live
        under `negative_particles/h_particle[_i]/{h, h_new}`; sigma is not a variable of the DBM graph"""
        out = {}
        for name, scope in self._var_names():
            base, i = name, 0
            if '_' in name and name.rsplit('_', 1)[1].isdigit():
                base, i = name.rsplit('_', 1)[0], int(name.rsplit('_', 1)[1])
            if name == 'sigma':
                tf_name = None
            elif base in ('h', 'h_new'):
                tf_name = 'negative_particles/h_particle%s/%s' % (self._sfx(i), base)
            else:
                tf_name = '%s/%s' % (scope, name)
            out[name] = (tf_name, self._engine.get(name))
        return out

    @classmethod
    def load_model(cls, model_path):
        model = super(DBM, cls).load_model(model_path)
        model.n_layers_ = len(model.n_hiddens_)
        return model

    # ---- schedules (reference dbm.py:771-791) -------------------------------------------
    def _feed(self, n_gibbs_steps=None):
        pick = lambda v: v[min(self.epoch_, len(v) - 1)]
        k = n_gibbs_steps if n_gibbs_steps is not None else pick(self.n_gibbs_steps)
        return float(pick(self.learning_rate)), float(pick(self.momentum)), int(k)

    def _check_batches(self, X, sharded=False):
        unit = self.batch_size * (getattr(self, '_world', 1) if sharded else 1)
        if len(X) % unit != 0:
            raise ValueError('DBM variational parameters have a fixed [batch_size, n] shape (dbm.py:345-348): '
                             '{0} rows are not a multiple of batch_size{2}={1}'
                             .format(len(X), unit, ' x world_size' if unit != self.batch_size else ''))

    # ---- training loop (reference dbm.py:793-857) -------------------------------------------
    def _train_epoch(self, Xd, N):
        lr, mom, k = self._feed()
        msres, nmfs = [], []
        if self._dp is not None:
            # data-parallel: the global minibatch is world * batch_size rows, this rank's slice starts at
            # rank * batch_size inside it (the msre summary is not fetched: it would be a rank-local number)
            for start in range(0, N, self.batch_size * self._world):
                self.iter_ += 1
                nmf = self._dp.train_step(Xd, lr, mom, k, row=start + self._rank * self.batch_size)
                if self.iter_ % self.train_metrics_every_iter == 0:
                    nmfs.append(nmf)
            return None, (np.mean(nmfs) if nmfs else None)
        for start in range(0, N, self.batch_size):
            self.iter_ += 1
            if self.iter_ % self.train_metrics_every_iter == 0:
                nmf, msre = self._engine.train_step(Xd, lr, mom, k, row=start, want_msre=True)
                msres.append(msre)
                nmfs.append(nmf)
            else:
                self._engine.train_step(Xd, lr, mom, k, row=start)
        return (np.mean(msres) if msres else None, np.mean(nmfs) if nmfs else None)

    def _run_val_metrics(self, X_val, Xvd):
        # one fetch of [msre, n_mf_updates] per batch (reference dbm.py:810-816).  Both tensors sit under the
        # control dependencies of the train graph (dbm.py:521-523): the fetch also advances the fantasy
        # particles by n_gibbs_steps, exactly like the reference's validation pass does.
        msres, nmfs = [], []
        _, _, k = self._feed()
        world, rank = getattr(self, '_world', 1), getattr(self, '_rank', 0)
        # (data-parallel: each rank evaluates its slice of every world*batch_size block; msre is then this
        # rank's share, n_mf_updates is global because the mean-field loop condition is)
        for start in range(rank * self.batch_size, len(X_val), self.batch_size * world):
            nmf, msre = self._engine.metrics(Xvd, k, row=start)
            msres.append(msre)
            nmfs.append(nmf)
        return np.mean(msres), np.mean(nmfs)

    def _fit(self, X, X_val=None, *args, **kwargs):
        dt = self._engine.dtype
        X = np.ascontiguousarray(X, dtype=dt)
        self._check_batches(X, sharded=True)
        Xd, N = as_device(X, dt), len(X)
        Xvd = None
        if X_val is not None:
            X_val = np.ascontiguousarray(X_val, dtype=dt)
            self._check_batches(X_val, sharded=True)
            Xvd = as_device(X_val, dt)
        val_msre, val_n_mf_updates = None, None
        for self.epoch_ in epoch_iter(start_epoch=self.epoch_, max_epoch=self.max_epoch, verbose=self.verbose):
            train_msre, train_n_mf_updates = self._train_epoch(Xd, N)
            if X_val is not None and self.epoch_ % self.val_metrics_every_epoch == 0:
                val_msre, val_n_mf_updates = self._run_val_metrics(X_val, Xvd)
            self._log_scalars('train', self.iter_, dict(msre=train_msre, n_mf_updates=train_n_mf_updates, epoch=self.epoch_))
            if X_val is not None and self.epoch_ % self.val_metrics_every_epoch == 0:
                self._log_scalars('val', self.iter_, dict(msre=val_msre, n_mf_updates=val_n_mf_updates))
            if self.verbose:        # the reference's progress line (dbm.py:843-854)
                shown = (('msre', train_msre, '%.5f'), ('n_mf_upds', train_n_mf_updates, '%.1f'),
                         ('val.msre', val_msre, '%.5f'), ('val.n_mf_upds', val_n_mf_updates, '%.1f'))
                write_during_training('; '.join(['epoch: %*d/%d' % (len(str(self.max_epoch)), self.epoch_, self.max_epoch)] +
                                                [('%s: ' + f) % (n, v) for n, v, f in shown if v]))
            self._display_dumps()
            if self.save_after_each_epoch:
                self._save_model(global_step=self.epoch_)
        self._engine.sync()

    def _display_dumps(self):
        """`display_filters` / `display_particles` (dbm.py:312-322, :531-547) as .npy files per epoch under logs/train:
        `W_filters_<i>` [n, h, w, c]: the first n columns of W_0 W_1 ... W_i as images (the reference's product and
        transposes); `particles_v` [n, h, w, c] and `particles_h_<i>` [n_particles, n]: the first n fantasy particles
        as they stand after the epoch (the reference shows the means of one more, unsampled, particle sweep)."""
        if self.display_filters:
            W = None
            for i in range(self.n_layers_):
                Wi = np.asarray(self._engine.get('W' + self._sfx(i)), dtype=np.float64)
                W = Wi if W is None else W.dot(Wi)
                if W.shape[0] == int(np.prod(self.v_shape)):
                    self._dump_array('W_filters_%d' % i, self._as_images(W.T[:self.display_filters]))
        if self.display_particles:
            n = self.display_particles
            v = self._engine.get('v')
            if v.shape[1] == int(np.prod(self.v_shape)):
                self._dump_array('particles_v', self._as_images(v[:n]))
            for i in range(self.n_layers_):
                self._dump_array('particles_h_%d' % i, self._engine.get('h' + self._sfx(i))[:, :n])

    # ---- public inference API ---------------------------------------------------------------
    # The reference re-loads the variables from disk at the start of every public call
    # (tf_model.py:22-28), so a call that does not save leaves the persistent state (the
    # mean-field parameters that seed the next minibatch, the fantasy particles) untouched.
    # Here the state is resident, so those calls snapshot and restore what they overwrite.
    def _snapshot(self, bases):
        names = [b + self._sfx(i) for b in bases for i in range(self.n_layers_) if b not in ('v', 'v_new')]
        names += [b for b in bases if b in ('v', 'v_new')]
        return {n: self._engine.get(n) for n in names}

    def _restore(self, snap):
        for n, a in snap.items():
            self._engine.set(n, a)

    @run_on_engine()
    def transform(self, X, np_dtype=None):
        """Mean-field activations of the last hidden layer (reference dbm.py:859-872)."""
        np_dtype = np_dtype or self._np_dtype
        dt = self._engine.dtype
        X = np.ascontiguousarray(X, dtype=dt)
        self._check_batches(X)
        Xd = as_device(X, dt)
        Gd = _ffi.DeviceArray((len(X), self.n_hiddens_[-1]), dt)
        snap = self._snapshot(('mu', 'mu_new'))
        for start in range(0, len(X), self.batch_size):
            self._engine.mean_field(Xd, row=start, out=Gd, out_row=start)
        self._engine.sync()
        self._restore(snap)
        return Gd.numpy().astype(np_dtype)

    @run_on_engine(update_seed=True)
    def reconstruct(self, X):
        """p(v | h_0 = q), q = mean-field p(h_0 | v = x) (reference dbm.py:874-885)."""
        dt = self._engine.dtype
        X = np.ascontiguousarray(X, dtype=dt)
        self._check_batches(X)
        Xd = as_device(X, dt)
        Rd = _ffi.DeviceArray(X.shape, dt)
        snap = self._snapshot(('mu', 'mu_new'))
        for start in range(0, len(X), self.batch_size):
            self._engine.reconstruct(Xd, Rd, row=start, out_row=start)
        self._engine.sync()
        self._restore(snap)
        return Rd.numpy()

    @run_on_engine(update_seed=True)
    def sample_v(self, n_gibbs_steps=0, save_model=False):
        """Visible particle activation probabilities after `n_gibbs_steps` sweeps
        (reference dbm.py:887-897, op :641-648)."""
        Vd = _ffi.DeviceArray((self.n_particles, self.n_visible_), self._engine.dtype)
        snap = None if save_model else self._snapshot(('v', 'v_new', 'h', 'h_new'))
        self._engine.sample_v(int(n_gibbs_steps), Vd)
        v = Vd.numpy()
        if save_model:
            self.n_samples_generated_ += n_gibbs_steps
            self._save_model()
        else:
            self._restore(snap)
        return v

    def set_ais_accumulation(self, dtype='float64'):
        """How `log_Z` accumulates the AIS log-weights.  'float64' (default): per chain the difference of consecutive
        log p*, row sums and running sum in double, fixed order - deterministic and closer to the exactly enumerable
        log Z.  'float32': the reference graph's ORDER of float32 accumulation (dbm.py:650-660, :708-728) - every
        log p*_beta(x) formed and added / subtracted in float32 in the graph's order (the reference's README notes the
        nats this loses at many betas); the row sums inside one log p* are the engine's slot partials, not
        TensorFlow's reduce_sum / einsum order, so this is the reference's arithmetic up to float32 round-off of
        those sums, not bit for bit.  Both agree with the reference to the 1e-5 the parity bar asks for at the beta counts tested."""
        if dtype not in ('float32', 'float64'):
            raise ValueError("dtype must be 'float32' or 'float64'")
        self._ais_literal = dtype == 'float32'
        return self

    def set_mean_field_arithmetic(self, mode='engine'):
        """Which sigmoid the Bernoulli layers evaluate (not a keyword of the reference: it has only its own).
        'engine' (default): one correctly rounded division, `e / (1 + e)` for x < 0 (~1.4 ulp).
        'reference': literally `tf.nn.sigmoid` as TensorFlow 1.3 computes it in float32, `1 / (1 + exp(-x))` with Eigen's
        exp (layers.py:47-48; ~1.8 ulp), in every pass of the engine.
        Parameters, means and metrics agree between the modes to float32 round-off.  What differs is the number of executed
        mean-field sweeps (`n_mf_updates` of the progress line): at the default mf_tol = 1e-7 the loop of reference
        dbm.py:449-452 is decided in the last bits of the means, and only with the reference's own sigmoid does the engine
        execute the reference's sweeps (784-512-1024, batch 512: 5-6 per update; 7-8 in the 'engine' arithmetic)."""
        if mode not in ('engine', 'reference'):
            raise ValueError("mode must be 'engine' or 'reference'")
        self._sigmoid_literal = mode == 'reference'
        if getattr(self, '_engine', None) is not None:
            self._engine.set_sigmoid_literal(self._sigmoid_literal)
        return self

    @run_on_engine(update_seed=True)
    def log_Z(self, n_betas=100, n_runs=100, n_gibbs_steps=5):
        """AIS estimate of the log partition function of the 2-layer binary DBM
        (reference dbm.py:899-939).  Returns log_mean, (log_low, log_high), values."""
        assert self.n_layers_ == 2
        assert self._all_bernoulli()           # reference dbm.py:926-927: every layer is a BernoulliLayer
        self._engine.set_ais_literal(self._ais_literal)
        if getattr(self, '_comm', None) is not None:
            # multi-GPU job: the independent chains are sharded over the ranks (their index in the RNG stream is
            # global), ONE all-gather of the per-chain values at the end; every rank returns all of them
            values = self._engine.ais_sharded(self._comm, n_betas, n_runs, n_gibbs_steps, seed=self._graph_seed)
        else:
            values = self._engine.ais(n_betas, n_runs, n_gibbs_steps, seed=self._graph_seed)
        log_mean = log_mean_exp(values)
        log_std = log_std_exp(values, log_mean_exp_x=log_mean)
        log_high = log_sum_exp([log_std, log_mean])
        log_low = log_diff_exp([log_std, log_mean])[0]
        return log_mean, (log_low, log_high), values

    @run_on_engine()
    def log_proba(self, X_test, log_Z):
        """Variational lower bound on log p(x) for the 2-layer binary DBM (reference dbm.py:941-957)."""
        assert self.n_layers_ == 2
        assert self._all_bernoulli()           # reference dbm.py:947-948
        X_test = np.ascontiguousarray(X_test, dtype=self._engine.dtype)
        self._check_batches(X_test)
        Xd = as_device(X_test, self._engine.dtype)
        P = np.zeros(len(X_test))
        snap = self._snapshot(('mu', 'mu_new'))
        for start in range(0, len(X_test), self.batch_size):
            P[start:start + self.batch_size] = self._engine.log_proba(Xd, row=start)
        self._restore(snap)
        return P - log_Z
