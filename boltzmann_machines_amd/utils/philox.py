"""Host-side (numpy) Philox4x32-10 in TensorFlow's stream convention.

Used by the model classes for the initialisers the reference evaluates once on
the host side of `session.run` — `tf.random_normal(stddev=W_init, seed=...)`
(rbm/base_rbm.py:277-279), `layer.init` for DBM particles (layers.py:43-45,
dbm.py:362-383) — and by bench.py for its seeded synthetic inputs.  Same
generator as csrc/bm_rng.h; the reference's own known-answer
test (rbm/tests/test_rbm.py:64-67) pins it: seed pair (87654321, 1337),
stddev 0.01 -> W[0][0] = -0.0094548017 (f32) / -0.0077341544416 (f64).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
DEFAULT_GRAPH_SEED = 87654321          # TF: op-level seed only -> (DEFAULT_GRAPH_SEED, op_seed)
_MASK32 = np.uint64(0xFFFFFFFF)


def philox_blocks(seed, site, call, block0, nblocks):
    """words [nblocks, 4] (uint32) of counters (block, site, call) under key `seed`."""
    blk = np.arange(block0, block0 + nblocks, dtype=np.uint64)
    c0 = (blk & _MASK32).astype(np.uint32)
    c1 = (blk >> np.uint64(32)).astype(np.uint32)
    c2 = np.full(nblocks, site, dtype=np.uint32)
    c3 = np.full(nblocks, call, dtype=np.uint32)
    k0 = np.uint32(int(seed) & 0xFFFFFFFF)
    k1 = np.uint32((int(seed) >> 32) & 0xFFFFFFFF)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK32).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return np.stack([c0, c1, c2, c3], axis=1)


def _u32_to_f32(x):
    return ((x & np.uint32(0x7FFFFF)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)


def _u32x2_to_f64(x0, x1):
    m = ((x0.astype(np.uint64) & np.uint64(0xFFFFF)) << np.uint64(32)) | x1.astype(np.uint64)
    return (m | (np.uint64(1023) << np.uint64(52))).view(np.float64) - 1.0


def uniform(seed, site, call, n, idx0=0, dtype=np.float32):
    """n uniforms in [0,1).  float32: element i = word i%4 of block i/4 (TF Uint32ToFloat); float64: element i = the word
    pair i%2 of block i/2 (TF Uint64ToDouble)."""
    if np.dtype(dtype) == np.float64:
        b0, b1 = idx0 // 2, (idx0 + n + 1) // 2
        w = philox_blocks(seed, site, call, b0, b1 - b0).reshape(-1, 2)      # rows = word pairs
        return _u32x2_to_f64(w[:, 0], w[:, 1])[idx0 - 2 * b0: idx0 - 2 * b0 + n]
    b0, b1 = idx0 // 4, (idx0 + n + 3) // 4
    w = philox_blocks(seed, site, call, b0, b1 - b0).reshape(-1)
    return _u32_to_f32(w[idx0 - 4 * b0: idx0 - 4 * b0 + n])


def normal(seed, site, call, n, dtype=np.float32):
    """TF random_normal stream: Box-Muller on word pairs (4 f32 or 2 f64 normals per block)."""
    if np.dtype(dtype) == np.float32:
        w = philox_blocks(seed, site, call, 0, (n + 3) // 4)
        u1 = np.maximum(_u32_to_f32(w[:, 0::2]), np.float32(1.0e-7))
        v1 = np.float32(2.0 * np.pi) * _u32_to_f32(w[:, 1::2])
        r = np.sqrt(np.float32(-2.0) * np.log(u1))
        out = np.stack([np.sin(v1) * r, np.cos(v1) * r], axis=2).astype(np.float32)   # [nb, 2 pairs, (sin,cos)]
        return out.reshape(-1)[:n]
    w = philox_blocks(seed, site, call, 0, (n + 1) // 2)
    u1 = np.maximum(_u32x2_to_f64(w[:, 0], w[:, 1]), 1.0e-20)
    v1 = 2.0 * np.pi * _u32x2_to_f64(w[:, 2], w[:, 3])
    r = np.sqrt(-2.0 * np.log(u1))
    return np.stack([np.sin(v1) * r, np.cos(v1) * r], axis=1).reshape(-1)[:n]


def tf_random_normal(shape, stddev, op_seed, dtype=np.float32):
    """tf.random_normal(shape, mean=0, stddev, seed=op_seed) with no graph seed set."""
    n = int(np.prod(shape))
    x = normal(DEFAULT_GRAPH_SEED, int(op_seed) & 0xFFFFFFFF, (int(op_seed) >> 32) & 0xFFFFFFFF, n, dtype)
    return (x * np.dtype(dtype).type(stddev)).reshape(shape)
