"""Scenarios of public-API calls that are executed, line for line, against BOTH packages:

  * the UNMODIFIED reference (`/root/reference/boltzmann_machines`) running on the NumPy TF-1 stand-in
    (tests/tf1_shim) - `tests/golden/make_golden_from_reference.py` records what it returns in
    `tests/golden/ref_<scenario>.npz`;
  * `boltzmann_machines_amd` - on the GPU (`tests/test_reference_fixtures_gpu.py`) and, with the C oracle standing in
    for the device engine, on the CPU (`tests/test_reference_fixtures.py`).

A scenario is a function `f(pkg, workdir) -> {name: ndarray}`; `pkg` offers BernoulliRBM, GaussianRBM,
MultinomialRBM, DBM and RNG with the reference's signatures.  Everything a scenario touches is the drop-in surface:
constructor keywords, fit / transform / init_from / set_params / load_model / get_tf_params, DBM.reconstruct /
sample_v / log_Z / log_proba, the progress lines (metrics), epoch_ / iter_ bookkeeping.  Test infrastructure."""
import contextlib
import io
import os
import re

import numpy as np

_FLOAT = r'[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|nan|inf)'


class Capture(object):
    """progress lines the models print per epoch (`epoch: 1/2; msre: ...`): the reference's only scalar outlet of
    its train / validation metrics besides TensorBoard"""

    def __init__(self):
        self.buf = io.StringIO()

    def __enter__(self):
        self._cm = contextlib.redirect_stdout(self.buf)
        self._cm.__enter__()
        return self

    def __exit__(self, *exc):
        return self._cm.__exit__(*exc)

    def metrics(self):
        """[n_epoch_lines, n_numbers] array of every number on the `epoch:` lines, and the key names"""
        rows, keys = [], None
        for line in self.buf.getvalue().replace('\r', '\n').split('\n'):
            line = line.strip()
            if not line.startswith('epoch:'):
                continue
            parts = [p.strip() for p in line.split(';')]
            k, v = [], []
            for p in parts[1:]:
                m = re.match(r'^([\w.]+)\s*:\s*(%s)$' % _FLOAT, p)
                if m:
                    k.append(m.group(1))
                    v.append(float(m.group(2)))
            rows.append(v)
            keys = k if keys is None or len(k) > len(keys) else keys
        width = max([len(r) for r in rows] + [0])
        out = np.full((len(rows), width), np.nan)
        for i, r in enumerate(rows):
            out[i, :len(r)] = r
        return out, keys or []


def _params(model, out, tag, stride=None):
    for k, v in model.get_tf_params().items():
        a = np.asarray(v)
        if stride and a.ndim == 2 and a.size > 20000:
            a = a[::stride[0], ::stride[1]]
        out['%s:%s' % (tag, k)] = a
    out['%s:epoch_iter' % tag] = np.array([model.epoch_, model.iter_])


def _metrics_cfg(**kw):
    d = dict(l2_loss=True, msre=True, pll=True, feg=True, l2_loss_fmt='.8e', msre_fmt='.8e', pll_fmt='.8e',
             feg_fmt='.8e', train_metrics_every_iter=1, val_metrics_every_epoch=1, feg_every_epoch=1,
             n_batches_for_feg=2)
    d.update(kw)
    return d


# ------------------------------------------------------------------------------------------------ RBM scenarios
def rbm_reference_test_config(pkg, d, cls_name='BernoulliRBM', dtype='float32', **extra):
    """the configuration of the reference's own test (rbm/tests/test_rbm.py:12-22, :69-114): real-valued inputs,
    dropout, both layers sampled, a short last batch; fit - transform - resume - load_model - resume, all metrics on"""
    X = pkg.RNG(seed=1337).rand(16, 12)
    X_val = pkg.RNG(seed=42).rand(8, 12)
    cfg = dict(n_visible=12, n_hidden=8, sample_v_states=True, sample_h_states=True, dropout=0.9, verbose=True,
               display_filters=False, random_seed=1337, dtype=dtype, max_epoch=2, metrics_config=_metrics_cfg(),
               model_path=os.path.join(d, 'm/'))
    cfg.update(extra)
    C = getattr(pkg, cls_name)
    out = {}
    cap = Capture()
    with cap:
        rbm = C(**cfg)
        rbm.fit(X, X_val)
    _params(rbm, out, 'fit2')
    with cap:
        out['transform2'] = rbm.transform(X_val)
        rbm.set_params(max_epoch=3).fit(X, X_val)
    _params(rbm, out, 'fit3')
    rbm = C.load_model(os.path.join(d, 'm/'))
    _params(rbm, out, 'loaded')
    with cap:
        out['transform3'] = rbm.transform(X_val)
        rbm.set_params(max_epoch=4).fit(X)
    _params(rbm, out, 'fit4')
    out['metrics'], _ = cap.metrics()
    return out


def rbm_float64(pkg, d):
    return rbm_reference_test_config(pkg, d, dtype='float64')


def rbm_multinomial(pkg, d):
    return rbm_reference_test_config(pkg, d, cls_name='MultinomialRBM', n_samples=7)


def rbm_gaussian(pkg, d):
    """GaussianRBM with a per-unit sigma (rbm.py:88-116): input / sigma once, reconstructions `x * sigma + b` not
    re-divided (layers.py:84-86); Normal sampling of the visibles"""
    sigma = list(np.linspace(0.6, 1.4, 12))
    return rbm_reference_test_config(pkg, d, cls_name='GaussianRBM', sigma=sigma, learning_rate=0.01, dropout=None)


def rbm_schedules(pkg, d):
    """per-epoch schedules (1-based index, base_rbm.py:535-541), the tf.while_loop chain (n_gibbs_steps is a list),
    sparsity penalty, momentum / learning-rate lists, binary data, no dropout, init_from"""
    V, H, N, bs = 20, 12, 37, 10
    X = (pkg.RNG(seed=5).rand(N, V) < 0.3).astype(np.float32)
    X_val = (pkg.RNG(seed=6).rand(14, V) < 0.3).astype(np.float32)
    kw = dict(n_visible=V, n_hidden=H, batch_size=bs, max_epoch=3, learning_rate=[0.3, 0.05, 0.02],
              momentum=[0.1, 0.5, 0.9], n_gibbs_steps=[4, 1, 2], l2=1e-3, sample_v_states=True, random_seed=77,
              verbose=True, sparsity_cost=0.01, sparsity_target=0.2, metrics_config=_metrics_cfg(train_metrics_every_iter=3))
    out = {}
    cap = Capture()
    with cap:
        rbm = pkg.BernoulliRBM(model_path=os.path.join(d, 'a/'), **kw)
        rbm.fit(X, X_val)
    _params(rbm, out, 'fit3')
    with cap:
        out['transform'] = rbm.transform(X_val)
        rbm2 = pkg.BernoulliRBM(model_path=os.path.join(d, 'b/'), **dict(kw, max_epoch=4, random_seed=78))
        rbm2.init_from(rbm)
        # init_from copies EVERY trailing-underscore attribute (base_rbm.py:682-685), `initialized_` included, and the
        # reference's next fit() would then import a meta graph that was never written for the new model path
        # (tf_model.py:22-23): a warm-started model trains only after the flag is cleared
        rbm2.initialized_ = False
        rbm2.fit(X)
    _params(rbm2, out, 'init_from_fit4')
    out['metrics'], _ = cap.metrics()
    return out


def rbm_means_only(pkg, d):
    """neither layer sampled inside the chain (sample_h_states=False: h0 means feed the chain, base_rbm.py:422-424),
    CD-3 unrolled (scalar n_gibbs_steps), dbm_first / dbm_last multipliers (base_rbm.py:256-262)"""
    V, H, N = 16, 10, 30
    X = pkg.RNG(seed=9).rand(N, V).astype(np.float32)
    out = {}
    for tag, kw in (('first', dict(dbm_first=True)), ('last', dict(dbm_last=True, sample_h_states=False)),
                    ('both_sampled', dict(dbm_first=True, dbm_last=True, sample_v_states=True))):
        rbm = pkg.BernoulliRBM(n_visible=V, n_hidden=H, batch_size=8, max_epoch=2, n_gibbs_steps=3, learning_rate=0.1,
                               random_seed=21, verbose=False, model_path=os.path.join(d, tag + '/'), **kw)
        rbm.fit(X)
        _params(rbm, out, tag)
        out[tag + ':transform'] = rbm.transform(X[:8])
    return out


def rbm_config0_shape(pkg, d):
    """BASELINE configs[0]: BernoulliRBM 784 x 128, CD-1, batch 100 (examples/rbm_mnist.py: lr 0.05, momentum
    0.5 -> 0.9, l2 1e-5, hidden states sampled, visible means) on 100 synthetic binary rows, 2 epochs; matrices stored
    with a stride.  (Kept this short on purpose: a run with ~10^6 Bernoulli draws always contains one within 1e-6 of
    a tie, where the float32 round-off of the matmul order decides the sample.)"""
    V, H, N = 784, 128, 100
    X = (pkg.RNG(seed=50).rand(N, V) < 0.1307).astype(np.float32)
    rbm = pkg.BernoulliRBM(n_visible=V, n_hidden=H, batch_size=100, max_epoch=2, learning_rate=0.05,
                           momentum=[0.5, 0.5, 0.9], l2=1e-5, W_init=0.01, random_seed=1337, verbose=False,
                           model_path=os.path.join(d, 'm/'))
    rbm.fit(X)
    out = {}
    _params(rbm, out, 'fit2', stride=(7, 3))
    out['transform'] = rbm.transform(X[:100])[::5, ::3]
    return out


# ------------------------------------------------------------------------------------------------ DBM scenarios
def _pretrain(pkg, d, X, sizes, v_cls='BernoulliRBM', top_cls='BernoulliRBM', epochs=2, seeds=(11, 12, 13), dtype=None, **top_kw):
    rbms, Q = [], X
    for i in range(len(sizes) - 1):
        last = i == len(sizes) - 2
        C = getattr(pkg, v_cls if i == 0 else (top_cls if last else 'BernoulliRBM'))
        kw = dict(n_visible=sizes[i], n_hidden=sizes[i + 1], dbm_first=(i == 0 and len(sizes) > 2),
                  dbm_last=(last and len(sizes) > 2), max_epoch=epochs, batch_size=10, learning_rate=0.05,
                  random_seed=seeds[i], verbose=False, model_path=os.path.join(d, 'rbm%d/' % i))
        if dtype:
            kw['dtype'] = dtype
        if last:
            kw.update(top_kw)
        if i == 0 and v_cls == 'GaussianRBM':
            kw.update(learning_rate=0.005, sigma=1.)
        r = C(**kw)
        r.fit(Q)
        rbms.append(r)
        Q = r.transform(Q)
    return rbms, Q


def _dbm_walk(pkg, d, dbm, X, X_val, out, ais=True):
    """the public inference calls of dbm.py:859-957 in a fixed order, then a resumed fit"""
    cap = Capture()
    with cap:
        dbm.fit(X, X_val)
    _params(dbm, out, 'fit2')
    with cap:
        out['transform'] = dbm.transform(X_val)
        out['reconstruct'] = dbm.reconstruct(X_val)
        out['sample_v_0'] = dbm.sample_v(n_gibbs_steps=0)
        out['sample_v_3'] = dbm.sample_v(n_gibbs_steps=3)
        out['sample_v_2_saved'] = dbm.sample_v(n_gibbs_steps=2, save_model=True)
    out['n_samples_generated'] = np.array([dbm.n_samples_generated_])
    _params(dbm, out, 'after_sample_v')
    if ais:
        with cap:
            log_mean, (log_low, log_high), values = dbm.log_Z(n_betas=40, n_runs=6, n_gibbs_steps=2)
            out['log_Z'] = np.array([log_mean, log_low, log_high])
            out['log_Z_values'] = np.asarray(values)
            out['log_proba'] = dbm.log_proba(X_val, log_Z=log_mean)
    with cap:
        dbm.set_params(max_epoch=3).fit(X)
    _params(dbm, out, 'fit3')
    dbm2 = pkg.DBM.load_model(os.path.join(d, 'dbm/'))
    with cap:
        out['loaded:transform'] = dbm2.transform(X_val)
        dbm2.set_params(max_epoch=4).fit(X)
    _params(dbm2, out, 'loaded_fit4')
    m, keys = cap.metrics()
    # the executed mean-field trip counts separately: with mf_tol at the float32 round-off of the residual (1e-7, the
    # reference's default) the sweep at which `max |mu - mu_new| > tol` turns false depends on the last bit
    n_mf = [i for i, k in enumerate(keys) if 'n_mf' in k]
    out['metrics'] = m[:, [i for i in range(m.shape[1]) if i not in n_mf]]
    out['metrics_n_mf_updates'] = m[:, n_mf]
    return out


def dbm_two_layers(pkg, d):
    """20-12-16 DBM from two pre-trained RBMs (dbm.py:266-291), mean-field + PCD with a per-epoch sweep schedule,
    max-norm, per-layer sparsity (incl. the q_means[i] scalar-index quirk, dbm.py:581-590), validation fetches that
    advance the particles (dbm.py:810-816 under :521-523), all inference calls, AIS, ELBO, resume, load_model"""
    V, N = 20, 40
    X = (pkg.RNG(seed=5).rand(N, V) < 0.3).astype(np.float32)
    X_val = (pkg.RNG(seed=6).rand(20, V) < 0.3).astype(np.float32)
    rbms, _ = _pretrain(pkg, d, X, (V, 12, 16))
    dbm = pkg.DBM(rbms=rbms, n_particles=10, batch_size=10, n_gibbs_steps=[1, 2, 3], max_mf_updates=20, mf_tol=1e-5,
                  learning_rate=[0.02, 0.01], momentum=[0.5, 0.9], max_epoch=2, l2=1e-4, max_norm=0.6,
                  sparsity_cost=[0.01, 0.02], sparsity_target=[0.2, 0.1], random_seed=13, verbose=True,
                  train_metrics_every_iter=2, model_path=os.path.join(d, 'dbm/'))
    return _dbm_walk(pkg, d, dbm, X, X_val, {})


def dbm_float64(pkg, d):
    """the 20-12-16 walk of `dbm_two_layers` with dtype='float64' (base/mixin.py:14-25; the DBM graph is built "all in model
    dtype", dbm.py:294-383): float64 RBMs, a float64 DBM, mean-field at the reference's default tolerance, every public call"""
    V, N = 20, 40
    X = (pkg.RNG(seed=5).rand(N, V) < 0.3).astype(np.float64)
    X_val = (pkg.RNG(seed=6).rand(20, V) < 0.3).astype(np.float64)
    rbms, _ = _pretrain(pkg, d, X, (V, 12, 16), dtype='float64')
    dbm = pkg.DBM(rbms=rbms, n_particles=10, batch_size=10, n_gibbs_steps=[1, 2, 3], max_mf_updates=20, mf_tol=1e-7,
                  learning_rate=[0.02, 0.01], momentum=[0.5, 0.9], max_epoch=2, l2=1e-4, max_norm=0.6,
                  sparsity_cost=[0.01, 0.02], sparsity_target=[0.2, 0.1], random_seed=13, verbose=True,
                  train_metrics_every_iter=2, dtype='float64', model_path=os.path.join(d, 'dbm/'))
    return _dbm_walk(pkg, d, dbm, X, X_val, {})


def dbm_three_layers(pkg, d):
    """16-10-8-6: the intermediate RBM is halved (dbm.py:277-280), middle layers read the NEW layer below and the OLD
    layer above (dbm.py:400-407); particles initialised from data / features (examples/dbm_mnist.py:137-141);
    visible states not sampled"""
    V, N = 16, 30
    X = (pkg.RNG(seed=15).rand(N, V) < 0.4).astype(np.float32)
    X_val = (pkg.RNG(seed=16).rand(10, V) < 0.4).astype(np.float32)
    rbms, _ = _pretrain(pkg, d, X, (V, 10, 8, 6))
    Q = rbms[0].transform(X[:10])
    G = rbms[1].transform(Q)
    T = rbms[2].transform(G)
    dbm = pkg.DBM(rbms=rbms, n_particles=10, v_particle_init=X[:10].copy(), h_particles_init=(Q, G, T),
                  batch_size=10, n_gibbs_steps=2, max_mf_updates=8, mf_tol=1e-7, learning_rate=0.01, max_epoch=2,
                  l2=1e-5, sample_v_states=False, sample_h_states=(True, True, False), sparsity_cost=0.005,
                  random_seed=31, verbose=True, train_metrics_every_iter=1, model_path=os.path.join(d, 'dbm/'))
    return _dbm_walk(pkg, d, dbm, X, X_val, {}, ais=False)


def dbm_gaussian_bernoulli_multinomial(pkg, d):
    """the unit types of examples/dbm_cifar.py: Gaussian visible, Bernoulli middle, Multinomial top layer.
    (A ONE-layer DBM - the README's "DBM class can be used also for training RBM", README.md:96, i.e. the PCD form of
    BASELINE configs[2] - cannot be built by the reference as written: `_make_tf_model` always builds the AIS and ELBO
    graphs, which index W[1] / hb[1], dbm.py:674,757-768 -> IndexError.  The Gaussian-visible particle path is covered
    here instead; the engine's 1-layer path is checked against the oracle, tests/test_dbm_parity_gpu.py.)"""
    V, N = 12, 30
    X = pkg.RNG(seed=45).randn(N, V).astype(np.float32)
    X_val = pkg.RNG(seed=46).randn(10, V).astype(np.float32)
    rbms, _ = _pretrain(pkg, d, X, (V, 10, 6), v_cls='GaussianRBM', top_cls='MultinomialRBM', n_samples=5)
    dbm = pkg.DBM(rbms=rbms, n_particles=10, batch_size=10, n_gibbs_steps=2, max_mf_updates=6, learning_rate=0.002,
                  max_epoch=2, l2=1e-3, random_seed=51, verbose=True, train_metrics_every_iter=1,
                  model_path=os.path.join(d, 'dbm/'))
    return _dbm_walk(pkg, d, dbm, X, X_val, {}, ais=False)


# ------------------------------------------------------------------------- scenarios at the BASELINE.json sizes
# These are the shapes at which the device kernels leave their guarded edge path: K >= 512 runs the LDS-DMA steady
# loop of the tile engine, 512-row batches fill the tuned geometries, 5 PCD sweeps / 50 mean-field sweeps / 1000
# beta steps go through the chained launches.  With ~10^6 .. 10^8 Bernoulli draws per scenario some draws lie within
# float32 round-off of a tie; they are RECORDED (the `near_ties*` arrays of the fixture), not rejected - see
# tests/reference_fixtures.py for how the comparison uses them.  `pkg.mark(label)` tells the generator which public
# call the following draws belong to (a no-op for the package under test).
def _mark(pkg, label):
    getattr(pkg, 'mark', lambda s: None)(label)


def rbm_config1_shape(pkg, d, seed=1337):
    """BASELINE configs[1] = the workload bench.py times: BernoulliRBM 784 x 1024, CD-1, batch 512, both layers
    sampled, lr 0.05 / momentum 0.9 / l2 1e-5 (examples/rbm_mnist.py:160,166,55), W ~ N(0, 0.01^2): two updates on
    1024 synthetic Bernoulli(0.1307) rows through `_make_train_op` (rbm/base_rbm.py:415-479), then `transform` of
    one batch (base_rbm.py:687-700).  Matrices are stored with a stride."""
    V, H, B = 784, 1024, 512
    X = (pkg.RNG(seed=50).rand(2 * B, V) < 0.1307).astype(np.float32)
    _mark(pkg, 'fit')
    rbm = pkg.BernoulliRBM(n_visible=V, n_hidden=H, batch_size=B, max_epoch=1, learning_rate=0.05, momentum=0.9,
                           l2=1e-5, W_init=0.01, sample_v_states=True, sample_h_states=True, n_gibbs_steps=1,
                           random_seed=seed, verbose=False, model_path=os.path.join(d, 'm/'))
    rbm.fit(X)
    out = {}
    _params(rbm, out, 'fit1', stride=(7, 3))
    _mark(pkg, 'transform')
    out['transform'] = rbm.transform(X[:B])[:, ::8]
    return out


def _config3_stack(pkg, d, seeds=(1337, 1111), W_init=0.01):
    """784-512-1024 with the synthetic weights of SURVEY 8(d) cfg4: two RBMs initialised (`init()`, tf_model.py:168-173)
    with W ~ N(0, 0.01^2), not trained - the point of these scenarios is the DBM's own graphs at full size"""
    sizes = (784, 512, 1024)
    rbms = []
    for i in range(2):
        r = pkg.BernoulliRBM(n_visible=sizes[i], n_hidden=sizes[i + 1], dbm_first=(i == 0), dbm_last=(i == 1),
                             W_init=W_init, random_seed=seeds[i], verbose=False,
                             model_path=os.path.join(d, 'rbm%d/' % i))
        r.init()
        rbms.append(r)
    return rbms


def _dbm_config3(pkg, d, B, seed):
    """BASELINE configs[3] (one rank of it): mean-field (<= 50 sweeps, tol 1e-7) + PCD-5 with `B` particles, max-norm 6,
    both sparsity penalties, lr 2e-3, l2 1e-7 (examples/dbm_mnist.py:250-284); one epoch of two updates
    (dbm.py:515-639, :429-509), then `transform` (dbm.py:859-872) of one batch"""
    X = (pkg.RNG(seed=60).rand(2 * B, 784) < 0.1307).astype(np.float32)
    _mark(pkg, 'pretrain')
    rbms = _config3_stack(pkg, d)
    _mark(pkg, 'fit')
    cap = Capture()
    with cap:
        dbm = pkg.DBM(rbms=rbms, n_particles=B, batch_size=B, n_gibbs_steps=5, max_mf_updates=50, mf_tol=1e-7,
                      learning_rate=2e-3, momentum=0.9, max_epoch=1, l2=1e-7, max_norm=6.,
                      sparsity_target=(0.2, 0.1), sparsity_cost=(1e-4, 5e-5), sparsity_damping=0.9,
                      random_seed=seed, verbose=True, train_metrics_every_iter=1,
                      model_path=os.path.join(d, 'dbm/'))
        dbm.fit(X)
    out = {}
    _params(dbm, out, 'fit1', stride=(7, 3))
    _mark(pkg, 'transform')
    with cap:
        out['transform'] = dbm.transform(X[:B])[:, ::8]
    m, keys = cap.metrics()
    n_mf = [i for i, k in enumerate(keys) if 'n_mf' in k]
    out['metrics'] = m[:, [i for i in range(m.shape[1]) if i not in n_mf]]
    out['metrics_n_mf_updates'] = m[:, n_mf]
    return out


def dbm_config3_shape_b100(pkg, d, seed=2222):
    """batch_size = n_particles = 100, the example's own numbers (examples/dbm_mnist.py:250,267)"""
    return _dbm_config3(pkg, d, 100, seed)


def dbm_config3_shape_b512(pkg, d, seed=2222):
    """batch_size = n_particles = 512, the size bench.py runs configs[3] at"""
    return _dbm_config3(pkg, d, 512, seed)


def ais_config4_slice(pkg, d, seed=2222):
    """BASELINE configs[4], a 64-chain slice: AIS over 1000 betas on the 784-512-1024 stack, one Gibbs step per
    transition (dbm.py:696-736, :899-939).  ~1.5e8 Bernoulli draws: chains are independent, so a draw at a float32
    tie can fork ONE chain; `near_ties` names the chains that had such a draw.  Weights ~ N(0, 0.04^2), the scale of a
    trained model: with the N(0, 0.01^2) of an untrained stack the chains' log-weights agree to 5e-5 of their value
    whatever the trajectory, and the comparison would not notice a fork."""
    _mark(pkg, 'pretrain')
    rbms = _config3_stack(pkg, d, W_init=0.04)
    dbm = pkg.DBM(rbms=rbms, n_particles=10, batch_size=10, max_epoch=1, random_seed=seed, verbose=False,
                  model_path=os.path.join(d, 'dbm/'))
    dbm.init()
    _mark(pkg, 'ais')
    log_mean, (log_low, log_high), values = dbm.log_Z(n_betas=1000, n_runs=64, n_gibbs_steps=1)
    return {'log_Z': np.array([log_mean, log_low, log_high]), 'log_Z_values': np.asarray(values)}


SCENARIOS = dict((f.__name__, f) for f in (
    rbm_reference_test_config, rbm_float64, rbm_multinomial, rbm_gaussian, rbm_schedules, rbm_means_only,
    rbm_config0_shape, dbm_two_layers, dbm_float64, dbm_three_layers, dbm_gaussian_bernoulli_multinomial,
    rbm_config1_shape, dbm_config3_shape_b100, dbm_config3_shape_b512, ais_config4_slice))

# the scenarios at the BASELINE sizes (slower: seconds on the device, a minute or two on the CPU oracle)
FULL_SIZE = ('rbm_config1_shape', 'dbm_config3_shape_b100', 'dbm_config3_shape_b512', 'ais_config4_slice')

# outputs whose ROWS are independent trajectories (one minibatch row / one AIS chain each), with the label of the
# public call that produced them: a near-tie recorded for row r under that label can fork row r only
ROW_LOCAL = {
    'rbm_config1_shape': {'transform': 'transform'},
    'dbm_config3_shape_b100': {},
    'dbm_config3_shape_b512': {},
    'ais_config4_slice': {'log_Z_values': 'ais'},
}
# outputs that aggregate over the rows of a row-local output (compared only when no row forked)
ROW_AGGREGATE = {'ais_config4_slice': {'log_Z': 'log_Z_values'}}

# scenarios whose trajectories contain Normal draws (the stand-in's Box-Muller rounds every float32 operation once
# from float64, tests/tf1_shim/tensorflow/_philox.py, the device's is pinned operation by operation: both within
# half an ulp per operation of the exact value, and the scenarios hold north_star's 1e-5 like all others)
GAUSSIAN = {'rbm_gaussian', 'dbm_gaussian_bernoulli_multinomial'}

# Executed mean-field sweeps (dbm.py:449-452; `metrics_n_mf_updates`).  The DBM scenarios are compared in the engine's
# "reference arithmetic" (DBM.set_mean_field_arithmetic('reference') / BM355_SIGMOID_LITERAL=1: the literal float32 tf.sigmoid,
# bit-identical between kernel, oracle and stand-in) and the trip counts must then be the reference's - exactly, except for the
# single sweep of a loop the GENERATOR recorded as ending at the float32 noise floor of mf_tol (`mf_loops` of the fixture,
# tests/golden/make_golden_from_reference.py; the rule is tests/reference_fixtures.py mf_trip_bounds).  In the default
# arithmetic the loop runs a sweep or two longer at mf_tol = 1e-7 (the engine's sigmoid, one correctly rounded division, passes
# a one-ulp movement of a pre-activation on to the mean where the literal form's coarser steps for x < 0 absorb it):
# tests/test_reference_fixtures*.py state those counts separately.
DBM_SCENARIOS = ('dbm_two_layers', 'dbm_three_layers', 'dbm_gaussian_bernoulli_multinomial',
                 'dbm_config3_shape_b100', 'dbm_config3_shape_b512')
