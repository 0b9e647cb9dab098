import os
import sys

import pytest

# the CPU oracle is OpenMP: on the 256-thread GPU-box host the default team oversubscribes the
# tiny test problems (minutes of spin-waiting); a small passive team is faster there
os.environ.setdefault('OMP_NUM_THREADS', '16')
os.environ.setdefault('OMP_WAIT_POLICY', 'passive')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def gpu_lib():
    """libbm355 with a visible GPU; -m gpu tests FAIL (not skip) if the HIP path is unusable."""
    from boltzmann_machines_amd import _ffi
    lib = _ffi.load()
    assert lib.bm_device_count() > 0, 'no HIP device visible: -m gpu tests need the GPU box'
    return lib
