"""boltzmann_machines_amd — MI355X-native engine for the RBM/DBM hot path of
yell/boltzmann-machines.  The compute path is libbm355.so
(hand-written HIP for gfx950 behind the C-ABI of include/bm355.h); importing this
package never falls back to a CPU implementation."""
from .rbm import BernoulliRBM, MultinomialRBM, GaussianRBM, BaseRBM, logit_mean          # noqa: F401
from .base import EngineModel, BaseModel                                  # noqa: F401
from .utils import RNG                                                    # noqa: F401

try:                                                                      # DBM lands after the RBM path
    from .dbm import DBM                                                  # noqa: F401
except ImportError:                                                       # pragma: no cover
    pass

__version__ = '0.1'
