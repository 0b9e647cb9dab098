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


def _tols(name):
    if name == 'rbm_float64':
        return dict(rtol=1e-11, metrics_rtol=1e-7, atol=1e-15)
    if name in scenarios.GAUSSIAN:
        return dict(rtol=5e-5, metrics_rtol=5e-5)
    if name == 'dbm_three_layers':
        return dict(rtol=1e-5, metrics_rtol=1e-5, n_mf_atol=1.0)     # mf_tol at round-off level, see the CPU leg
    return dict(rtol=1e-5, metrics_rtol=1e-5)


@pytest.mark.parametrize('name', sorted(scenarios.SCENARIOS))
def test_hip_path_reproduces_the_reference_fixture(gpu_lib, name, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    got = scenarios.SCENARIOS[name](rf.OursPackage(), str(tmp_path))
    rf.compare(name, got, rf.load(name), **_tols(name))
