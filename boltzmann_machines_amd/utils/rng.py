"""Host MT19937 stream with a JSON-able state (reference utils/rng.py:4-62).

Every public model call draws its graph seed from this stream
(`SeedMixin.make_random_seed`, reference base/mixin.py:28-35) and the state is
checkpointed in random_state.json (reference base/tf_model.py:131-134,156-159).
"""
import numpy as np


class RNG(np.random.RandomState):
    """
    >>> rng = RNG(1337)
    >>> state = rng.get_state()
    >>> rng.rand()
    0.2620246750155817
    >>> rng.rand()
    0.1586839721544656
    >>> _ = rng.reseed()
    >>> rng.rand()
    0.2620246750155817
    >>> _ = rng.set_state(state)
    >>> rng.rand()
    0.2620246750155817
    """

    def __init__(self, seed=None):
        self._seed = seed
        super(RNG, self).__init__(self._seed)

    def reseed(self):
        if self._seed is not None:
            self.seed(self._seed)
        return self

    def get_state(self):
        state = list(super(RNG, self).get_state())
        state[1] = state[1].tolist()
        return state

    def set_state(self, state):
        state = list(state)
        state[1] = np.asarray(state[1], dtype=np.uint32)
        super(RNG, self).set_state(tuple(state))
        return self
