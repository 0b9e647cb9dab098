"""Port of the reference's own RBM tests (boltzmann_machines/rbm/tests/test_rbm.py)
to the MI355X classes — same shapes, seeds and assertions — plus checks that the
Python fit loop drives the engine exactly like the oracle twin driven by hand."""
import os
import shutil

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_almost_equal

from boltzmann_machines_amd import BernoulliRBM, GaussianRBM, MultinomialRBM
from boltzmann_machines_amd.utils import RNG

pytestmark = pytest.mark.gpu

N_VISIBLE, N_HIDDEN = 12, 8
X = RNG(seed=1337).rand(16, N_VISIBLE)
X_VAL = RNG(seed=42).rand(8, N_VISIBLE)
CONFIG = dict(n_visible=N_VISIBLE, n_hidden=N_HIDDEN, sample_v_states=True, sample_h_states=True,
              dropout=0.9, verbose=False, display_filters=False, random_seed=1337)


@pytest.fixture
def dirs(tmp_path):
    d1, d2 = str(tmp_path / 'test_rbm_1') + '/', str(tmp_path / 'test_rbm_2') + '/'
    yield d1, d2
    for d in (d1, d2):
        shutil.rmtree(d, ignore_errors=True)


def compare_weights(rbm1, rbm2):
    w1, w2 = rbm1.get_tf_params(scope='weights'), rbm2.get_tf_params(scope='weights')
    for k in ('W', 'hb', 'vb'):
        assert_allclose(w1[k], w2[k])


def compare_transforms(rbm1, rbm2):
    H1, H2 = rbm1.transform(X_VAL), rbm2.transform(X_VAL)
    assert H1.shape == (len(X_VAL), N_HIDDEN)
    assert H1.shape == H2.shape
    assert_allclose(H1, H2)


@pytest.mark.parametrize('C,dtype', [(BernoulliRBM, 'float32'), (BernoulliRBM, 'float64'), (MultinomialRBM, 'float32'),
                                     (GaussianRBM, 'float32'), (MultinomialRBM, 'float64'), (GaussianRBM, 'float64')])
def test_initialization(gpu_lib, dirs, C, dtype):
    """reference test_rbm.py:52-67 — the W-init known answer after init()."""
    rbm = C(max_epoch=2, model_path=dirs[0], dtype=dtype, **CONFIG)
    rbm.init()
    w = rbm.get_tf_params(scope='weights')['W'][0][0]
    assert_almost_equal(w, -0.0094548017 if dtype == 'float32' else -0.0077341544416)


@pytest.mark.parametrize('C,dtype', [(BernoulliRBM, 'float32'), (BernoulliRBM, 'float64'), (MultinomialRBM, 'float32'),
                                     (GaussianRBM, 'float32'), (MultinomialRBM, 'float64'), (GaussianRBM, 'float64')])
def test_consistency(gpu_lib, dirs, C, dtype):
    """reference test_rbm.py:69-114 — twin models stay identical through fit, +1 epoch,
    load_model from disk, +1 epoch (same class / dtype list as the reference)."""
    rbm1 = C(max_epoch=2, model_path=dirs[0], dtype=dtype, **CONFIG)
    rbm2 = C(max_epoch=2, model_path=dirs[1], dtype=dtype, **CONFIG)
    rbm1.fit(X); rbm2.fit(X)
    compare_weights(rbm1, rbm2); compare_transforms(rbm1, rbm2)
    rbm1.set_params(max_epoch=rbm1.max_epoch + 1).fit(X)
    rbm2.set_params(max_epoch=rbm2.max_epoch + 1).fit(X)
    compare_weights(rbm1, rbm2); compare_transforms(rbm1, rbm2)
    w_before = rbm1.get_tf_params(scope='weights')
    rbm1 = C.load_model(dirs[0])
    rbm2 = C.load_model(dirs[1])
    assert rbm1.epoch_ == 3 and rbm1.iter_ == 3 * 2
    assert_allclose(rbm1.get_tf_params(scope='weights')['W'], w_before['W'])      # resume == state before
    compare_weights(rbm1, rbm2); compare_transforms(rbm1, rbm2)
    rbm1.set_params(max_epoch=rbm1.max_epoch + 1).fit(X)
    rbm2.set_params(max_epoch=rbm2.max_epoch + 1).fit(X)
    compare_weights(rbm1, rbm2); compare_transforms(rbm1, rbm2)
    assert rbm1.epoch_ == 4


def test_consistency_val(gpu_lib, dirs):
    """reference test_rbm.py:116-131 — same with validation metrics enabled."""
    mc = dict(msre=True, pll=True, feg=True, l2_loss=True, train_metrics_every_iter=1, feg_every_epoch=1)
    rbm1 = BernoulliRBM(max_epoch=2, model_path=dirs[0], metrics_config=dict(mc), **CONFIG)
    rbm2 = BernoulliRBM(max_epoch=2, model_path=dirs[1], metrics_config=dict(mc), **CONFIG)
    rbm1.fit(X, X_VAL); rbm2.fit(X, X_VAL)
    compare_weights(rbm1, rbm2); compare_transforms(rbm1, rbm2)


def test_fit_matches_oracle_driven_by_hand(gpu_lib, dirs):
    """fit() == the oracle twin stepped with the same seeds, schedules and batches
    (1-based schedule index, short last batch, one graph seed per public call)."""
    from oracle import oracle as orc
    V, H, N, bs = 20, 12, 37, 10
    Xd = (RNG(seed=5).rand(N, V) < 0.3).astype(np.float32)
    kw = dict(n_visible=V, n_hidden=H, batch_size=bs, max_epoch=3, learning_rate=[0.3, 0.05, 0.02], momentum=[0.1, 0.5, 0.9],
              n_gibbs_steps=[4, 1, 2], l2=1e-3, sample_v_states=True, random_seed=77, verbose=False,
              sparsity_cost=0.01, model_path=dirs[0])
    rbm = BernoulliRBM(**kw)
    rbm.fit(Xd)
    twin = orc.OracleRBM(V, H, sample_v_states=True, l2=1e-3, sparsity_cost=0.01)
    host = RNG(seed=77)
    graph_seed = host.randint(2 ** 31 - 1)                   # the fit() call's seed (tf_model.py:20-21)
    from boltzmann_machines_amd.utils import philox
    twin.p['W'][...] = (philox.normal(graph_seed, 77, 0, V * H) * np.float32(0.01)).reshape(V, H)
    twin.set_seed(graph_seed)
    for epoch in (1, 2, 3):
        i = min(epoch, 2)
        for s in range(0, N, bs):
            twin.train_step(Xd[s:s + bs], [0.3, 0.05, 0.02][i], [0.1, 0.5, 0.9][i], [4, 1, 2][i])
    p = rbm.get_tf_params(scope='weights')
    for k in ('W', 'vb', 'hb'):
        assert np.array_equal(p[k].view(np.uint32), twin.p[k].view(np.uint32)), k
    assert rbm.epoch_ == 3 and rbm.iter_ == 12


def test_errors(gpu_lib, dirs):
    rbm = BernoulliRBM(n_visible=4, n_hidden=3, model_path=dirs[0], verbose=False)
    with pytest.raises(RuntimeError):
        rbm.transform(np.zeros((2, 4)))                      # before fit/init (tf_model.py:29-30)
    rbm.init()
    with pytest.raises(ValueError):
        rbm.set_params(nope=1)
    with pytest.raises(RuntimeError):
        GaussianRBM.load_model(dirs[0])                      # class mismatch (tf_model.py:149-150)


def test_scalar_logs_and_checkpoint_files(gpu_lib, dirs):
    """what lands on disk after fit(): params.json (+ class name), random_state.json, the variable
    checkpoint, and the scalar logs that stand in for the TF summaries."""
    import json
    mc = dict(msre=True, pll=True, l2_loss=True, train_metrics_every_iter=1)
    rbm = BernoulliRBM(max_epoch=2, model_path=dirs[0], metrics_config=mc, **CONFIG)
    rbm.fit(X, X_VAL)
    d = dirs[0]
    p = json.load(open(os.path.join(d, 'params.json')))
    assert p['__class_name__'] == 'BernoulliRBM' and p['epoch_'] == 2 and p['n_hidden'] == N_HIDDEN
    assert os.path.isfile(os.path.join(d, 'random_state.json')) and os.path.isfile(os.path.join(d, 'model.npz'))
    z = np.load(os.path.join(d, 'model.npz'))
    assert set(z.files) >= {'W', 'vb', 'hb', 'dW', 'dvb', 'dhb', 'q_means'}
    tr = [json.loads(l) for l in open(os.path.join(d, 'logs/train/scalars.jsonl'))]
    va = [json.loads(l) for l in open(os.path.join(d, 'logs/val/scalars.jsonl'))]
    assert len(tr) == 2 and len(va) == 2 and {'msre', 'pll', 'l2_loss', 'epoch', 'step'} <= set(tr[0])
    assert tr[-1]['step'] == rbm.iter_ and np.isfinite(tr[-1]['pll']) and tr[-1]['msre'] > 0


def test_display_dumps_do_not_change_the_run(gpu_lib, dirs):
    """`display_filters` / `display_hidden_activations` (base_rbm.py:300-306, :429-435): one .npy per epoch under
    logs/train in the reference's image layout, and the trained weights do not depend on the display options."""
    cfg = dict(CONFIG, v_shape=(4, 3))
    a = BernoulliRBM(max_epoch=2, model_path=dirs[0], **dict(cfg, display_filters=3, display_hidden_activations=5))
    b = BernoulliRBM(max_epoch=2, model_path=dirs[1], **cfg)
    a.fit(X); b.fit(X)
    compare_weights(a, b)
    d = os.path.join(dirs[0], 'logs/train')
    for ep in (1, 2):
        f = np.load(os.path.join(d, 'W_filters_epoch%04d.npy' % ep))
        h = np.load(os.path.join(d, 'hidden_activation_means_epoch%04d.npy' % ep))
        assert f.shape == (3, 4, 3, 1) and h.shape == (10, 5) and np.all((h > 0) & (h < 1))
    W = a.get_tf_params(scope='weights')['W']
    assert_allclose(f[:, :, :, 0].reshape(3, 12), W.T[:3], rtol=1e-6)
    assert not os.path.exists(os.path.join(dirs[1], 'logs/train', 'W_filters_epoch0001.npy'))


def test_staged_checkpoints_hold_epoch_end_states(gpu_lib, dirs, monkeypatch):
    """per-epoch checkpoints are staged on the device in stream order (bm_rbm_stage) and read back by the writer thread
    while the next epoch runs: every snapshot that reaches the disk is the state at AN epoch end (never a mix of two),
    the last one is the final state, and the run equals the one with host-side snapshots (BM355_STAGED_SAVE=0)."""
    import time
    from boltzmann_machines_amd import base
    written = []
    real = np.savez

    def slow_savez(path, **kw):          # a slow disk: several epochs pass per write, slots get reused under the writer
        time.sleep(0.02)
        written.append({k: np.array(v) for k, v in kw.items()})
        return real(path, **kw)
    monkeypatch.setattr(base.np, 'savez', slow_savez)
    runs = {}
    for tag, env, d in (('staged', '1', dirs[0]), ('host', '0', dirs[1])):
        monkeypatch.setenv('BM355_STAGED_SAVE', env)
        del written[:]
        rbm = BernoulliRBM(max_epoch=14, model_path=d, **CONFIG)
        states = []
        orig = rbm._save_model

        def spy(*a, _orig=orig, _rbm=rbm, _states=states, **k):
            _states.append(_rbm._engine.get('W').copy())      # the true epoch-end state (a host sync; the test only)
            return _orig(*a, **k)
        rbm._save_model = spy
        rbm.fit(X)
        runs[tag] = (rbm.get_tf_params(scope='weights')['W'].copy(), [w['W'] for w in written], states)
    for tag in ('staged', 'host'):
        final, snaps, states = runs[tag]
        assert len(snaps) >= 2
        for s_ in snaps:                                      # each file content is some epoch-end state, whole
            assert any(np.array_equal(s_, st) for st in states), tag
        assert np.array_equal(snaps[-1], final), tag
    assert np.array_equal(runs['staged'][0], runs['host'][0])
    with np.load(os.path.join(dirs[0], 'model.npz')) as z:
        assert np.array_equal(z['W'], runs['staged'][0])
