"""CPU leg of the reference-fixture parity: the HOST LOGIC of boltzmann_machines_amd (schedules, seeds, MT stream,
call counters, batching, snapshot / restore, checkpoints, resume, the get_tf_params surface) with the C oracle in
the place of the device engine (tests/oracle_engine.py) must reproduce what the UNMODIFIED reference returned when
it ran on the TF-1 stand-in (tests/golden/ref_*.npz; generator: tests/golden/make_golden_from_reference.py).
The GPU leg - the same scenarios on libbm355 - is tests/test_reference_fixtures_gpu.py."""
import numpy as np
import pytest

from tests import oracle_engine, reference_fixtures as rf, reference_shim
from tests.golden import scenarios


def _tols(name):
    if name == 'rbm_float64':
        return dict(rtol=1e-11, metrics_rtol=1e-7, atol=1e-15)      # progress lines carry 9 significant digits
    if name in scenarios.GAUSSIAN:
        return dict(rtol=5e-5, metrics_rtol=5e-5)        # Box-Muller transcendentals: NumPy vs libm, last-bit differences
    if name == 'dbm_three_layers':
        # mf_tol = 1e-7 (the reference's default) sits at the float32 round-off of the residual: the sweep at which
        # the loop condition turns false may differ by one between two summation orders; mu itself agrees to 1e-5
        return dict(rtol=1e-5, metrics_rtol=1e-5, n_mf_atol=1.0)
    return dict(rtol=1e-5, metrics_rtol=1e-5)


@pytest.mark.parametrize('name', sorted(scenarios.SCENARIOS))
def test_host_logic_on_the_oracle_reproduces_the_reference_fixture(name, monkeypatch, tmp_path):
    oracle_engine.install(monkeypatch)
    monkeypatch.chdir(tmp_path)
    got = scenarios.SCENARIOS[name](rf.OursPackage(), str(tmp_path))
    rf.compare(name, got, rf.load(name), **_tols(name))


@pytest.mark.skipif(not reference_shim.available(), reason='the reference checkout is not on this box')
@pytest.mark.parametrize('name', ['rbm_reference_test_config', 'rbm_schedules', 'dbm_two_layers'])
def test_committed_fixture_is_what_the_reference_produces(name, tmp_path, monkeypatch):
    """regenerates a fixture from /root/reference on the TF-1 stand-in and compares it with the committed file"""
    from tests.golden import make_golden_from_reference as gen
    monkeypatch.chdir(tmp_path)
    out = gen.generate(name)
    ref = rf.load(name)
    assert sorted(out) == sorted(ref)
    for k in ref:
        assert np.allclose(out[k], ref[k], rtol=1e-6, atol=1e-9, equal_nan=True), k
