// bm_numerics.h — the scalar functions applied in the GEMM epilogues.
//
// sigmoid is specified operation-by-operation (DESIGN.md "Numerics"): only
// IEEE-754 correctly rounded fp32 ops (mul, add, fma, div, round-to-nearest-even)
// and integer bit operations, so the CPU oracle reproduces the probabilities
// BIT-FOR-BIT and `u < p` gives identical sample bitmaps.  The whole library is
// built with -ffp-contract=off; every fused multiply-add is an explicit fmaf.
//
// Reference ops restated: tf.nn.sigmoid (layers.py:47-48), tf.nn.softplus
// (rbm.py:20,114; dbm.py:656,659), tf.log_sigmoid (base_rbm.py:512).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace bm {

// exp(-a) for a in [0, 80], relative error < 1e-7
__device__ __forceinline__ float exp_neg(float a) {
    const float t = a * -1.44269504088896341f;     // -a*log2(e)
    const float n = rintf(t);                      // round-half-even
    float r = fmaf(n, -0.693145751953125f, -a);    // Cody-Waite, ln2_hi
    r = fmaf(n, -1.42860682030941723212e-6f, r);   // ln2_lo
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    const int ni = (int)n;                         // exact, n in [-116, 0]
    return __uint_as_float(__float_as_uint(p) + ((unsigned)ni << 23));
}

__device__ __forceinline__ float sigmoid(float x) {
    float a = fabsf(x);
    if (a > 80.0f) a = 80.0f;
    const float e = exp_neg(a);
    const float d = 1.0f + e;
    const float num = (x >= 0.0f) ? 1.0f : e;
    return num / d;                                // ONE correctly rounded IEEE division
}

// metrics only (tolerance-checked, not bit-pinned)
__device__ __forceinline__ float softplus(float x) {
    return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float log_sigmoid(float x) { return -softplus(-x); }

}  // namespace bm
