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

// "Reference arithmetic" (bm_dbm_set_sigmoid_literal): tf.nn.sigmoid as TensorFlow 1.3 evaluates it on the CPU,
//     y = 1 / (1 + exp(-x))        in float32 (Eigen scalar_sigmoid_op: pdiv(one, padd(one, pexp(pnegate(x)))))
// with exp = Eigen 3.3's pexp<Packet4f> (the Cephes expf scheme; the TF 1.3 wheels are SSE builds without FMA, so every
// pmadd is a rounded multiply followed by a rounded add).  TF and Eigen are not in /root/reference (pip dependency
// `tensorflow-gpu~=1.3.0`, requirements.txt:11): the algorithm is restated from Eigen/src/Core/arch/SSE/MathFunctions.h,
// operation by operation, so that oracle/bm_oracle.c (orc_exp_eigen) and the stand-in's tf.sigmoid (tests/tf1_shim) produce
// the same bits.  The point of the mode: the mean-field loop (dbm.py:449-452) ends when no mean moves by more than
// mf_tol = 1e-7, i.e. it is decided in the last bits of the sigmoid; with the literal form the engine executes the sweeps the
// reference's graph executes (5 - 6 per update at 784-512-1024, where the default form above runs 7 - 8).  ~1.8 ulp against
// the default's ~1.4.  The library is built with -ffp-contract=off: nothing below fuses.
__device__ __forceinline__ float exp_eigen(float x0) {
    float x = fminf(x0, 88.3762626647950f);
    x = fmaxf(x, -88.3762626647949f);
    float fx = x * 1.44269504088896341f;
    fx = fx + 0.5f;
    fx = floorf(fx);
    const float tmp = fx * 0.693359375f;
    const float zz = fx * -2.12194440e-4f;
    x = x - tmp;
    x = x - zz;
    const float z = x * x;
    float y = 1.9875691500E-4f;
    y = y * x; y = y + 1.3981999507E-3f;
    y = y * x; y = y + 8.3334519073E-3f;
    y = y * x; y = y + 4.1665795894E-2f;
    y = y * x; y = y + 1.6666665459E-1f;
    y = y * x; y = y + 5.0000001201E-1f;
    y = y * z; y = y + x;
    y = y + 1.0f;
    const float p2 = __uint_as_float((unsigned)((int)fx + 127) << 23);      // 2^fx, fx in [-127, 127]
    return fmaxf(y * p2, x0);
}
__device__ __forceinline__ float sigmoid_literal(float x) { return 1.0f / (1.0f + exp_eigen(-x)); }

// metrics only (tolerance-checked, not bit-pinned): softplus(x) = max(x, 0) + log1p(exp(-|x|)).
// The AIS log-weight epilogue evaluates it twice per output and was VALU-bound on the libm calls
// (expf + log1pf, ~100 instructions each): exp(-|x|) reuses exp_neg, and log1p(e), e in [0, 1], is
// 2*atanh(s) with s = e / (2 + e) <= 1/3 as an odd series to s^15 (truncation < 2e-8 relative).
// Measured against double precision: max relative error 2.5e-7 over [-79, 90] (absolute error < 2e-35 below).
__device__ __forceinline__ float log1p_unit(float e) {
    const float s = e / (2.0f + e);
    const float t = s * s;
    float p = 1.0f / 15.0f;
    p = fmaf(p, t, 1.0f / 13.0f);
    p = fmaf(p, t, 1.0f / 11.0f);
    p = fmaf(p, t, 1.0f / 9.0f);
    p = fmaf(p, t, 1.0f / 7.0f);
    p = fmaf(p, t, 1.0f / 5.0f);
    p = fmaf(p, t, 1.0f / 3.0f);
    p = fmaf(p, t, 1.0f);
    return 2.0f * s * p;
}
__device__ __forceinline__ float softplus(float x) {
    float a = fabsf(x);
    if (a > 80.0f) a = 80.0f;                      // exp(-80) ~ 1.8e-35: contributes nothing in fp32
    return fmaxf(x, 0.0f) + log1p_unit(exp_neg(a));
}
__device__ __forceinline__ float log_sigmoid(float x) { return -softplus(-x); }

// The same functions on the hardware transcendental units (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1e-7 relative), a third of
// the instructions of the forms above.  sigmoid_hw: fast-binary mode only (bm_bf3.h; tolerance parity, never the default
// path - a sigmoid decides sampled bits).  softplus_hw: also the AIS log-weight terms of the DEFAULT path (act_epilogue,
// difference form): those are tolerance-checked sums that feed no state, two softplus per output element against 512 k of
// matrix work made the epilogue what bounds the AIS passes (round 6, same-box A/B: 242.3 -> 229.9 ms per 300 betas x 20 000
// chains), and against the float64-accumulating oracle the values move no further than with the polynomial form
// (tools/ais_accuracy.py; profiles/r6_ais_softplus.txt).
__device__ __forceinline__ float sigmoid_hw(float x) {
    const float e = __builtin_amdgcn_exp2f(x * -1.44269504088896341f);      // exp(-x): inf for very negative x -> 0
    return __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float softplus_hw(float x) {
    const float e = __builtin_amdgcn_exp2f(fabsf(x) * -1.44269504088896341f);
    return fmaxf(x, 0.0f) + 0.693147180559945309f * __builtin_amdgcn_logf(1.0f + e);
}

}  // namespace bm
