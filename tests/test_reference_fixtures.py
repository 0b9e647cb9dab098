"""CPU leg of the reference-fixture parity: the HOST LOGIC of boltzmann_machines_amd (schedules, seeds, MT stream,
call counters, batching, snapshot / restore, checkpoints, resume, the get_tf_params surface) with the C oracle in
the place of the device engine (tests/oracle_engine.py) must reproduce what the UNMODIFIED reference returned when
it ran on the TF-1 stand-in (tests/golden/ref_*.npz; generator: tests/golden/make_golden_from_reference.py).
The GPU leg - the same scenarios on libbm355 - is tests/test_reference_fixtures_gpu.py."""
import numpy as np
import pytest

from tests import oracle_engine, reference_fixtures as rf, reference_shim
from tests.golden import scenarios


@pytest.mark.parametrize('name', sorted(scenarios.SCENARIOS))
def test_host_logic_on_the_oracle_reproduces_the_reference_fixture(name, monkeypatch, tmp_path):
    """DBM scenarios run in the engine's reference arithmetic (the literal float32 tf.sigmoid, layers.py:47-48): every output
    within 1e-5 AND the executed mean-field sweeps (`metrics_n_mf_updates`, dbm.py:449-452) the reference's own - exactly,
    or within the single sweep of a loop the generator recorded at the float32 noise floor of mf_tol (rf.mf_trip_bounds)"""
    oracle_engine.install(monkeypatch)
    monkeypatch.chdir(tmp_path)
    if name in scenarios.DBM_SCENARIOS:
        monkeypatch.setenv('BM355_SIGMOID_LITERAL', '1')
    got = scenarios.SCENARIOS[name](rf.OursPackage(), str(tmp_path))
    report = rf.check(name, got)
    print('\n'.join(['', name] + report))


@pytest.mark.parametrize('name,sweeps', [('dbm_three_layers', None), ('dbm_config3_shape_b100', (6.0, 8.0))])
def test_default_sigmoid_holds_the_fixture_and_states_its_own_trip_counts(name, sweeps, monkeypatch, tmp_path):
    """The DEFAULT arithmetic (the engine's one-division sigmoid, csrc/bm_numerics.h) on the same fixtures: every real output
    within the same 1e-5; the mean-field trip counts are the engine's own - never more than two sweeps from the reference's,
    and 6 to 8 sweeps per update at 784-512-1024 with mf_tol = 1e-7, where the reference (literal sigmoid) runs 5 to 6."""
    oracle_engine.install(monkeypatch)
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv('BM355_SIGMOID_LITERAL', raising=False)
    got = scenarios.SCENARIOS[name](rf.OursPackage(), str(tmp_path))
    ref = rf.load(name)
    report = rf.compare(name, got, ref, n_mf_atol=2.0, row_local=scenarios.ROW_LOCAL.get(name),
                        row_aggregate=scenarios.ROW_AGGREGATE.get(name), **rf.tolerances(name))
    print('\n'.join(['', name + ' (default sigmoid)'] + report))
    if sweeps:
        n = np.asarray(got['metrics_n_mf_updates']).ravel()
        assert np.all((n >= sweeps[0]) & (n <= sweeps[1])), n


def test_ais_slice_in_the_reference_order_of_float32_accumulation(monkeypatch, tmp_path):
    """CPU twin of the GPU test of the same name: the oracle's float32 accumulation mode against the reference's values"""
    oracle_engine.install(monkeypatch)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv('BM355_AIS_LITERAL', '1')
    got = scenarios.SCENARIOS['ais_config4_slice'](rf.OursPackage(), str(tmp_path))
    print('\n'.join(['', 'ais_config4_slice, float32 accumulation'] + rf.check('ais_config4_slice', got)))


@pytest.mark.skipif(not reference_shim.available(), reason='the reference checkout is not on this box')
@pytest.mark.parametrize('name', ['rbm_reference_test_config', 'rbm_schedules', 'dbm_two_layers', 'dbm_float64', 'rbm_config1_shape',
                                  'dbm_config3_shape_b100'])
def test_committed_fixture_is_what_the_reference_produces(name, tmp_path, monkeypatch):
    """regenerates a fixture from /root/reference on the TF-1 stand-in and compares it with the committed file"""
    from tests.golden import make_golden_from_reference as gen
    monkeypatch.chdir(tmp_path)
    out = gen.generate(name)
    ref = rf.load(name)
    assert sorted(out) == sorted(ref)
    for k in ref:
        if np.asarray(ref[k]).dtype.kind in 'US':
            assert [str(x) for x in out[k]] == [str(x) for x in ref[k]], k
        else:
            assert np.allclose(out[k], ref[k], rtol=1e-6, atol=1e-9, equal_nan=True), k
