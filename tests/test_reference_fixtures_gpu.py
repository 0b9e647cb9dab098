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
    monkeypatch.chdir(tmp_path)
    got = scenarios.SCENARIOS[name](rf.OursPackage(), str(tmp_path))
    report = rf.check(name, got)
    print('\n'.join(['', name] + report))


def test_ais_slice_in_the_reference_order_of_float32_accumulation(gpu_lib, tmp_path, monkeypatch):
    """the same 64 chains x 1000 betas with every log p*_beta(x) formed, added and subtracted in float32 in the order of
    the reference graph (dbm.py:650-660, :708-728; `DBM.set_ais_accumulation('float32')`) instead of the default
    per-chain double accumulation: both must hold the fixture's 1e-5; the report line shows the gap of each"""
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv('BM355_AIS_LITERAL', '1')
    got = scenarios.SCENARIOS['ais_config4_slice'](rf.OursPackage(), str(tmp_path))
    print('\n'.join(['', 'ais_config4_slice, float32 accumulation'] + rf.check('ais_config4_slice', got)))
