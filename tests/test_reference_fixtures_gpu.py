"""GPU leg of the reference-fixture parity: the scenarios of tests/golden/scenarios.py run through the public classes
of boltzmann_machines_amd ON libbm355 (HIP kernels through the C-ABI) and must reproduce what the UNMODIFIED
reference returned when it executed the same calls on the TF-1 stand-in (tests/golden/ref_*.npz; generator
tests/golden/make_golden_from_reference.py, which needs /root/reference and therefore runs in the build container
only - nothing here reads the reference).  Tolerances: north_star's 1e-5 float32 relative (plus one ulp of the
unit-scale intermediates); the sample bitmaps inside the trajectories are identical or the reals could not agree."""
import pytest

from tests import reference_fixtures as rf
from tests.golden import scenarios

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', sorted(scenarios.SCENARIOS))
def test_hip_path_reproduces_the_reference_fixture(gpu_lib, name, tmp_path, monkeypatch):
    """DBM scenarios run in the engine's reference arithmetic (bm_dbm_set_sigmoid_literal: the literal float32 tf.sigmoid of
    layers.py:47-48 in every epilogue): all outputs within 1e-5 AND the executed mean-field sweeps (`metrics_n_mf_updates`,
    dbm.py:449-452) the reference's own - exactly, or within the single sweep of a loop the generator recorded at the float32
    noise floor of mf_tol (rf.mf_trip_bounds)"""
    monkeypatch.chdir(tmp_path)
    if name in scenarios.DBM_SCENARIOS:
        monkeypatch.setenv('BM355_SIGMOID_LITERAL', '1')
    got = scenarios.SCENARIOS[name](rf.OursPackage(), str(tmp_path))
    report = rf.check(name, got)
    print('\n'.join(['', name] + report))


@pytest.mark.parametrize('name,sweeps', [('dbm_three_layers', None), ('dbm_gaussian_bernoulli_multinomial', None),
                                         ('dbm_config3_shape_b100', (6.0, 8.0)), ('dbm_config3_shape_b512', (6.0, 8.0))])
def test_default_sigmoid_holds_the_fixture_and_states_its_own_trip_counts(gpu_lib, name, sweeps, tmp_path, monkeypatch):
    """The DEFAULT arithmetic (the engine's one-division sigmoid, csrc/bm_numerics.h) on the same fixtures: every real output
    within the same 1e-5; the mean-field trip counts are the engine's own - never more than two sweeps from the reference's,
    and 6 to 8 sweeps per update at 784-512-1024 with mf_tol = 1e-7, where the reference (literal sigmoid) runs 5 to 6."""
    import numpy as np
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


def test_ais_slice_in_the_reference_order_of_float32_accumulation(gpu_lib, tmp_path, monkeypatch):
    """the same 64 chains x 1000 betas with every log p*_beta(x) formed, added and subtracted in float32 in the order of
    the reference graph (dbm.py:650-660, :708-728; `DBM.set_ais_accumulation('float32')`) instead of the default
    per-chain double accumulation: both must hold the fixture's 1e-5; the report line shows the gap of each"""
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv('BM355_AIS_LITERAL', '1')
    got = scenarios.SCENARIOS['ais_config4_slice'](rf.OursPackage(), str(tmp_path))
    print('\n'.join(['', 'ais_config4_slice, float32 accumulation'] + rf.check('ais_config4_slice', got)))
