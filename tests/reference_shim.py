"""Import the UNMODIFIED reference package (/root/reference/boltzmann_machines) on top of the NumPy TF-1 stand-in
(tests/tf1_shim).  Test infrastructure: used by tests/golden/make_golden_from_reference.py and
tests/test_reference_shim.py, in THIS container only (the reference does not exist on the GPU box)."""
import os
import sys

REFERENCE = os.environ.get('BM_REFERENCE_ROOT', '/root/reference')
SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tf1_shim')


def available():
    return os.path.isdir(os.path.join(REFERENCE, 'boltzmann_machines'))


class _LayersAlias(object):
    """rbm/rbm.py imports `layers` as a TOP-LEVEL module (rbm/env.py puts the package directory on sys.path) while
    dbm.py imports `.layers`: two module objects, so `isinstance(layer, BernoulliLayer)` in DBM.log_Z (dbm.py:925-927)
    fails whichever way the reference is imported.  Resolving the top-level name to the package's module (an import
    alias, no reference file is touched) makes both names one module."""

    @staticmethod
    def find_spec(name, path=None, target=None):
        if name == 'layers' and 'boltzmann_machines.layers' in sys.modules:
            import importlib.machinery
            mod = sys.modules['boltzmann_machines.layers']

            class _Loader(object):
                @staticmethod
                def create_module(spec):
                    return mod

                @staticmethod
                def exec_module(module):
                    pass
            return importlib.machinery.ModuleSpec('layers', _Loader())
        return None


def activate():
    """returns (tensorflow shim module, reference `boltzmann_machines` package)"""
    if not available():
        raise RuntimeError('reference checkout not found at %s' % REFERENCE)
    os.environ.setdefault('MPLBACKEND', 'Agg')
    for p in (os.path.join(REFERENCE, 'boltzmann_machines'), REFERENCE, SHIM):   # rbm/env.py does the same for `layers`
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    if not any(isinstance(f, type) and f.__name__ == '_LayersAlias' for f in sys.meta_path):
        sys.meta_path.insert(0, _LayersAlias)
    import tensorflow as tf
    assert 'numpy-shim' in tf.__version__, 'a real TensorFlow shadows the shim'
    import tensorflow.contrib.distributions  # noqa: F401
    import boltzmann_machines as bm
    assert os.path.abspath(bm.__file__).startswith(os.path.abspath(REFERENCE)), bm.__file__
    return tf, bm
