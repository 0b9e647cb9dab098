"""The reference's OWN tests, run against the unmodified reference package on the NumPy TF-1 stand-in
(tests/tf1_shim): if these pass, the stand-in carries the reference's graph builders, session / Saver round trips,
`load_model` and resume correctly - which is what entitles the fixtures it generates (tests/golden/ref_*.npz) to be
called reference output.  Runs where /root/reference exists (the build container); skipped on the GPU box."""
import os
import sys

import pytest

from tests import reference_shim

pytestmark = pytest.mark.skipif(not reference_shim.available(), reason='the reference checkout is not on this box')


@pytest.fixture()
def reference_tests(tmp_path, monkeypatch):
    tf, bm = reference_shim.activate()
    from tests.golden import reference_rng_policy
    tf.set_rng_policy(reference_rng_policy.policy)
    import boltzmann_machines.rbm
    import boltzmann_machines.utils
    # rbm/tests/test_rbm.py imports `rbm` and `utils` as top-level packages (nose ran it from inside the package)
    monkeypatch.setitem(sys.modules, 'rbm', boltzmann_machines.rbm)
    monkeypatch.setitem(sys.modules, 'utils', boltzmann_machines.utils)
    monkeypatch.syspath_prepend(os.path.join(reference_shim.REFERENCE, 'boltzmann_machines', 'rbm', 'tests'))
    monkeypatch.chdir(tmp_path)                       # the tests write test_rbm_1/ and test_rbm_2/ into the cwd
    sys.modules.pop('test_rbm', None)
    import test_rbm
    return test_rbm.TestRBM()


@pytest.mark.parametrize('case', ['test_W_init', 'test_initialization', 'test_consistency', 'test_consistency_val'])
def test_reference_test_rbm(reference_tests, case):
    """rbm/tests/test_rbm.py:29-131: constructor errors, the W-init known answers through the reference's `init()`
    (-0.0094548017 float32 / -0.0077341544416 float64), determinism + resume + load_model for BernoulliRBM
    (float32, float64), MultinomialRBM, GaussianRBM"""
    try:
        getattr(reference_tests, case)()
    finally:
        reference_tests.cleanup()


def test_reference_path_table(monkeypatch):
    """base/tests/test_tf_model.py: the working-path table of TensorFlowModel.compute_working_paths"""
    reference_shim.activate()
    monkeypatch.syspath_prepend(os.path.join(reference_shim.REFERENCE, 'boltzmann_machines', 'base', 'tests'))
    sys.modules.pop('test_tf_model', None)
    import test_tf_model
    t, ran = test_tf_model.TestWorkingPaths(), 0
    for name in dir(t):
        if name.startswith('test'):
            getattr(t, name)()
            ran += 1
    assert ran >= 3


def test_reference_ais_is_the_literal_float32_accumulation(tmp_path, monkeypatch):
    """The reference's AIS graph (dbm.py:696-736) run on the stand-in - float32 `log_Z +=` / `-=` by construction -
    against the oracle on the same parameters and chains: the oracle's LITERAL mode follows it to float32 round-off;
    the default double accumulation stays within the 1e-5 parity bar at this length."""
    import numpy as np
    from tests.golden import make_golden_from_reference as gen, scenarios
    from oracle import oracle as orc
    monkeypatch.chdir(tmp_path)
    pkg = gen.ReferencePackage()
    V = 20
    X = (pkg.RNG(seed=5).rand(40, V) < 0.3).astype(np.float32)
    rbms, _ = scenarios._pretrain(pkg, str(tmp_path), X, (V, 12, 16))
    dbm = pkg.DBM(rbms=rbms, n_particles=10, batch_size=10, max_epoch=1, random_seed=13, verbose=False,
                  model_path=str(tmp_path / 'dbm') + '/')
    dbm.init()
    seeds = []
    orig = pkg.tf.set_random_seed
    monkeypatch.setattr(pkg.tf, 'set_random_seed', lambda s: (seeds.append(s), orig(s))[1])
    _, _, values = dbm.log_Z(n_betas=600, n_runs=12, n_gibbs_steps=1)
    p = dbm.get_tf_params(scope='weights')
    twin = orc.OracleDBM(V, [12, 16], n_particles=10, batch_size=10)
    for k in ('W', 'W_1', 'vb', 'hb', 'hb_1'):
        twin.p[k][...] = p[k]
    lit = twin.ais(600, 12, 1, seeds[-1], literal=True)
    dbl = twin.ais(600, 12, 1, seeds[-1])
    np.testing.assert_allclose(lit, values, rtol=2e-6)
    np.testing.assert_allclose(dbl, values, rtol=1e-5)


def test_literal_sigmoid_is_the_stand_ins_tf_sigmoid_bit_for_bit():
    """`tf.sigmoid` as float32 `1 / (1 + exp(-x))` (layers.py:47-48): Eigen's scalar_sigmoid_op over Eigen's pexp in TF 1.3,
    restated operation by operation in the stand-in (tests/tf1_shim `_exp_eigen_f32`).  The engine's reference arithmetic
    (orc_sigmoid_literal = csrc/bm_numerics.h sigmoid_literal) is that function BIT FOR BIT - which is what lets the fixtures
    compare the executed mean-field sweeps (dbm.py:449-452; tests/reference_fixtures.py mf_trip_bounds).  The engine's
    DEFAULT sigmoid (`e / (1 + e)` for x < 0, one division: orc_sigmoid) differs from it in the last bit and is the more
    accurate of the two.

    (Round 5's stand-in evaluated the same formula with NumPy's float32 exp: that combination steps by 1.19e-7 > mf_tol
    between consecutive arguments, kept the stand-in's loop running to max_mf_updates = 50, and was reported as the
    reference's behaviour.  With Eigen's exp restated no step exceeds 5.96e-8 and the stand-in runs 5 - 6 sweeps at
    784-512-1024: the 50 was NumPy's, not the reference's.  Asserted below so the finding stays checked.)"""
    import numpy as np
    tf, _ = reference_shim.activate()
    from oracle import oracle as orc
    step = dict(lit=0.0, ours=0.0, npexp=0.0)
    err = dict(lit=0.0, ours=0.0)
    for start in (-0.9, -0.6, -0.4, -0.24, -0.1, 0.1, 0.3, 0.7, -30.0, -8.0, 8.0, 30.0):
        x0 = np.float32(start)
        x = x0 + np.arange(3000, dtype=np.float32) * np.spacing(x0)          # consecutive float32 arguments
        ref = np.asarray(tf._sigmoid(x), dtype=np.float32)
        ours = np.array([orc.lib().orc_sigmoid(float(v)) for v in x], dtype=np.float32)
        lit = np.array([orc.lib().orc_sigmoid_literal(float(v)) for v in x], dtype=np.float32)
        assert np.array_equal(lit.view(np.uint32), ref.view(np.uint32)), start
        if abs(start) > 1:
            continue
        npexp = (np.float32(1) / (np.float32(1) + np.exp(-x))).astype(np.float32)
        exact = 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
        for k, v in (('lit', lit), ('ours', ours), ('npexp', npexp)):
            step[k] = max(step[k], float(np.abs(np.diff(v)).max()))
        err['lit'] = max(err['lit'], float((np.abs(lit - exact) / np.spacing(lit)).max()))
        err['ours'] = max(err['ours'], float((np.abs(ours - exact) / np.spacing(ours)).max()))
    assert step['lit'] <= 6e-8 and step['ours'] <= 6e-8, step      # neither form steps by more than mf_tol = 1e-7 ...
    assert err['ours'] < err['lit'] < 2.0, err                     # (~1.4 against ~1.8 ulp)
    if step['npexp'] <= 1e-7:                                      # ... NumPy's exp in the same formula did (NumPy-version dependent)
        pytest.skip('this NumPy\'s float32 exp no longer shows the 1.19e-7 step')
