"""Philox4x32-10 in TensorFlow's stream convention, NumPy (part of the TF-1 shim: test infrastructure).

Written from the published algorithm (Salmon et al., SC'11) and TF's `random_distributions.h` conventions; pinned by
the reference's own known answer (rbm/tests/test_rbm.py:64-67: seed pair (87654321, 1337), stddev 0.01 ->
W[0][0] = -0.0094548017 float32, -0.0077341544416 float64), which tests/test_reference_shim.py re-checks through
the reference's `init()`."""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_LO = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def blocks(key, w2, w3, block0, nblocks):
    """[nblocks, 4] uint32 output words of counters (block_lo, block_hi, w2, w3) under the 64-bit key"""
    blk = np.arange(block0, block0 + nblocks, dtype=np.uint64)
    c0, c1 = (blk & _LO).astype(np.uint32), (blk >> _S32).astype(np.uint32)
    c2 = np.full(nblocks, int(w2) & 0xFFFFFFFF, dtype=np.uint32)
    c3 = np.full(nblocks, int(w3) & 0xFFFFFFFF, dtype=np.uint32)
    k0, k1 = int(key) & 0xFFFFFFFF, (int(key) >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c0.astype(np.uint64), _M1 * c2.astype(np.uint64)
        hi0, lo0 = (p0 >> _S32).astype(np.uint32), (p0 & _LO).astype(np.uint32)
        hi1, lo1 = (p1 >> _S32).astype(np.uint32), (p1 & _LO).astype(np.uint32)
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint32(k0), lo1, hi0 ^ c3 ^ np.uint32(k1), lo0
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=1)


def words(key, w2, w3, n, idx0=0):
    b0, b1 = idx0 // 4, (idx0 + n + 3) // 4
    return blocks(key, w2, w3, b0, b1 - b0).reshape(-1)[idx0 - 4 * b0: idx0 - 4 * b0 + n]


def _f32(x):                                    # Uint32ToFloat: 23 mantissa bits, [1, 2) - 1
    return ((x & np.uint32(0x7FFFFF)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)


def _f64(x0, x1):                               # Uint64ToDouble: 52 mantissa bits from a word pair
    m = ((x0.astype(np.uint64) & np.uint64(0xFFFFF)) << _S32) | x1.astype(np.uint64)
    return (m | (np.uint64(1023) << np.uint64(52))).view(np.float64) - 1.0


def uniform(key, w2, w3, n, dtype=np.float32, idx0=0):
    if np.dtype(dtype) == np.float32:
        return _f32(words(key, w2, w3, n, idx0))
    b0, b1 = idx0 // 2, (idx0 + n + 1) // 2     # two doubles per block
    w = blocks(key, w2, w3, b0, b1 - b0)
    u = np.stack([_f64(w[:, 0], w[:, 1]), _f64(w[:, 2], w[:, 3])], axis=1).reshape(-1)
    return u[idx0 - 2 * b0: idx0 - 2 * b0 + n]


def normal(key, w2, w3, n, dtype=np.float32):
    """Box-Muller on word pairs: (sin, cos)(2 pi u2) * sqrt(-2 ln u1); 4 float32 or 2 float64 normals per block"""
    if np.dtype(dtype) == np.float32:
        # TF's BoxMullerFloat takes float32 log / sqrt / sincos from the platform's libm; which last bit those return
        # is not part of TF's contract.  The stand-in evaluates each float32 operation of that function in float64 and
        # rounds it ONCE (= a correctly rounded libm), so that it sits within half an ulp per operation of any
        # implementation instead of adding NumPy's own float32 SIMD error (up to 1.4 ulp in exp / log / sin).
        w = blocks(key, w2, w3, 0, (n + 3) // 4)
        u1 = np.maximum(_f32(w[:, 0::2]), np.float32(1.0e-7))
        v1 = np.float32(2.0 * np.pi) * _f32(w[:, 1::2])
        lg = np.log(u1.astype(np.float64)).astype(np.float32)
        r = np.sqrt((np.float32(-2.0) * lg).astype(np.float64)).astype(np.float32)
        s = np.sin(v1.astype(np.float64)).astype(np.float32)
        c = np.cos(v1.astype(np.float64)).astype(np.float32)
        return np.stack([s * r, c * r], axis=2).astype(np.float32).reshape(-1)[:n]
    w = blocks(key, w2, w3, 0, (n + 1) // 2)
    u1 = np.maximum(_f64(w[:, 0], w[:, 1]), 1.0e-20)
    v1 = 2.0 * np.pi * _f64(w[:, 2], w[:, 3])
    r = np.sqrt(-2.0 * np.log(u1))
    return np.stack([np.sin(v1) * r, np.cos(v1) * r], axis=1).reshape(-1)[:n]
