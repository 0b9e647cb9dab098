"""Host-side Mersenne Twister stream whose state round-trips through JSON.

The drop-in contract this file carries (reference utils/rng.py:4-62, base/mixin.py:28-35, base/tf_model.py:131-134,
156-159): `RNG(seed)` IS a `numpy.random.RandomState` seeded the way the reference seeds it, so that
`make_random_seed()` hands every public call the reference's seed sequence; `get_state()` returns the five MT19937
fields as plain Python values - the schema of `random_state.json` - and `set_state()` takes that list (or NumPy's own
tuple) back; `reseed()` returns the stream to its state after construction.
"""
import numpy as np

_Base = np.random.RandomState


class RNG(_Base):
    """
    >>> rng = RNG(1337)
    >>> saved = rng.get_state()
    >>> rng.rand()
    0.2620246750155817
    >>> rng.rand()
    0.1586839721544656
    >>> rng.reseed().rand()
    0.2620246750155817
    >>> import json
    >>> rng.set_state(json.loads(json.dumps(saved))).rand()
    0.2620246750155817
    """

    def __init__(self, seed=None):
        _Base.__init__(self, seed)
        self._seed = seed

    def reseed(self):
        """back to the state right after construction; a stream built without a seed has nothing to return to"""
        if self._seed is None:
            return self
        _Base.seed(self, self._seed)
        return self

    def get_state(self):
        kind, key, pos, has_gauss, cached_gaussian = _Base.get_state(self)
        return [kind, [int(word) for word in key], int(pos), int(has_gauss), float(cached_gaussian)]

    def set_state(self, state):
        kind, key, pos, has_gauss, cached_gaussian = state
        _Base.set_state(self, (kind, np.array(key, dtype=np.uint32), int(pos), int(has_gauss), float(cached_gaussian)))
        return self
