"""tf.contrib.distributions.{Bernoulli, Normal, Multinomial} of TF 1.3 as the reference uses them
(layers.py:34-36,50-51,67-69,88-89, rbm/rbm.py:56, dbm.py:701).  Part of the TF-1 shim: test infrastructure.

  * Bernoulli(probs | logits).sample(seed) = cast(random_uniform(shape(probs), dtype=probs.dtype, seed) < probs, int32)
    (TF 1.3 `Bernoulli._sample_n`);
  * Normal(loc, scale).sample(seed) = random_normal(shape, 0, 1, dtype=loc.dtype, seed) * scale + loc
    (TF 1.3 `Normal._sample_n`);
  * Multinomial(total_count, probs | logits).sample() = per batch row the counts of `total_count` categorical draws.
    TF draws them with `tf.multinomial`, whose kernel consumes its Philox stream in an undocumented way; the stand-in
    is the engine's documented definition (oracle/bm_oracle.c `softmax_multinomial_row`): draw d of row r takes the
    uniform with flat index r * total_count + d of the op's stream, t = u * S with S the row's sequential float32
    cumulative sum c of the probabilities, category = smallest i with c[i] > t.
"""
import numpy as np

import tensorflow as tf
from tensorflow import _philox

_trace_margin = [None]
_tie_eps = [0.0]


def set_margin_trace(fn, tie_eps=0.0):
    """fn(scope, min |u - p| over the draw, near_ties, number of draws): lets the fixture generator COUNT the draws whose outcome the
    float32 round-off of another summation order could flip.  near_ties = [n, 3] float64 rows (row, column, u - p)
    of every draw with |u - p| < tie_eps (row = flat index / last dimension: the minibatch row / particle / chain)"""
    _trace_margin[0] = fn
    _tie_eps[0] = float(tie_eps)


def _k_bernoulli_less(ctx, t, u, p):
    if _trace_margin[0] is not None and np.size(u):
        d = np.asarray(u, dtype=np.float64) - np.asarray(p, dtype=np.float64)
        a = np.abs(d)
        ties = np.zeros((0, 3))
        if _tie_eps[0] > 0.0:
            idx = np.flatnonzero(a.reshape(-1) < _tie_eps[0])
            if idx.size:
                ncol = a.shape[-1] if a.ndim else 1
                ties = np.stack([idx // ncol, idx % ncol, d.reshape(-1)[idx]], axis=1).astype(np.float64)
        _trace_margin[0](t.scope, float(a.min()), ties, int(a.size))
    return np.less(u, p)


tf.register_kernel('bernoulli_less', _k_bernoulli_less)


class Bernoulli(object):
    def __init__(self, logits=None, probs=None, dtype=tf.int32, name='Bernoulli'):
        if (logits is None) == (probs is None):
            raise ValueError('Must pass probs or logits, but not both.')
        self.probs = tf.sigmoid(tf.convert_to_tensor(logits)) if probs is None else tf.convert_to_tensor(probs)
        self.dtype = dtype

    def sample(self, sample_shape=(), seed=None, name='sample'):
        assert sample_shape == ()
        with tf.name_scope('Bernoulli'):
            u = tf.random_uniform(tf.shape(self.probs), dtype=self.probs.dtype, seed=seed, role='Bernoulli.sample')
            less = tf.Tensor('bernoulli_less', [u, self.probs], dtype='bool')
            return tf.cast(less, self.dtype)


class Normal(object):
    def __init__(self, loc, scale, name='Normal'):
        self.loc = tf.convert_to_tensor(loc)
        self.scale = tf.convert_to_tensor(scale, self.loc.dtype)

    def sample(self, sample_shape=(), seed=None, name='sample'):
        assert sample_shape == ()
        with tf.name_scope('Normal'):
            z = tf.random_normal(tf.shape(self.loc + self.scale), mean=0., stddev=1., dtype=self.loc.dtype, seed=seed,
                                 role='Normal.sample')
            return z * self.scale + self.loc


def _k_multinomial(ctx, t, probs, total_count):
    probs = np.asarray(probs, dtype=np.float32)
    M = int(total_count)
    rows = probs.reshape(-1, probs.shape[-1])
    (key, w2, w3), site = tf._stream_of(t, ctx)
    ctx.used_rng = True
    u = _philox.uniform(key, w2, w3, rows.shape[0] * M, np.float32).reshape(rows.shape[0], M)
    out = np.zeros_like(rows)
    for r in range(rows.shape[0]):
        c = np.cumsum(rows[r], dtype=np.float32)            # sequential float32 prefix sum
        cat = np.searchsorted(c, u[r] * c[-1], side='right')  # smallest i with c[i] > t
        np.add.at(out[r], np.minimum(cat, len(c) - 1), np.float32(1))
    if tf._rng_trace[0] is not None:
        tf._rng_trace[0](site, (key, w2, w3), out)
    return out.reshape(probs.shape)


tf.register_kernel('multinomial', _k_multinomial)


class Multinomial(object):
    def __init__(self, total_count, logits=None, probs=None, name='Multinomial'):
        if (logits is None) == (probs is None):
            raise ValueError('Must pass probs or logits, but not both.')
        self.total_count = tf.convert_to_tensor(total_count, tf.float32)
        self.probs = tf.nn.softmax(tf.convert_to_tensor(logits)) if probs is None else tf.convert_to_tensor(probs)

    def sample(self, sample_shape=(), seed=None, name='sample'):
        assert sample_shape == ()
        with tf.name_scope('Multinomial'):
            return tf.Tensor('multinomial', [self.probs, self.total_count],
                             {'seed': seed, 'graph_seed': tf.get_default_graph().seed, 'role': 'Multinomial.sample'},
                             name='multinomial', dtype=tf.float32)
