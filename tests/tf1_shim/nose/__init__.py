"""stand-in for `nose` (the reference's utils/testing.py and base/tests import it; not installable offline)"""
from . import tools  # noqa: F401


def run(*a, **kw):
    raise RuntimeError('nose is not available: the reference tests are driven by tests/test_reference_shim.py')
