// bm_rng.h — Philox4x32-10 in TensorFlow's stream convention,
// the generator behind the reference's `Bernoulli(probs).sample()` /
// `tf.random_uniform` / `tf.random_normal` sites (layers.py:34-36,50-51,88-89;
// base_rbm.py:277-279,418,501).  Device-side, gfx950 only.
//
// Stream addressing used by every sampling site of this engine (DESIGN.md "RNG"):
//   key     = 64-bit seed of the public call        (k0 = lo, k1 = hi)
//   counter = (block_lo, block_hi, site, call)
//   block   = flat row-major element index / 4, word = index % 4
// so a [rows, cols] tensor sampled at (site, call) is independent of tiling,
// of the rank count (rows are GLOBAL rows) and reproducible on the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bm {

struct PhiloxKey {
    uint32_t k0, k1;   // seed lo / hi
    uint32_t site;     // counter word 2
    uint32_t call;     // counter word 3
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// the four words of element-block `block` of stream `key`
__device__ __forceinline__ void philox_block(const PhiloxKey &key, uint64_t block, uint32_t (&out)[4]) {
    philox4x32_10((uint32_t)block, (uint32_t)(block >> 32), key.site, key.call, key.k0, key.k1, out);
}

// Two consecutive Philox blocks (= a lane's 8 consecutive outputs).  act_kernel computes them
// in `fill()`, which the GEMM main loop calls while the first global loads are in flight
// (the wave idles for one memory round trip there), so the ~250 VALU ops cost nothing and
// stay out of the K loop, where every VALU instruction is paid in full (bm_gemm.h).
struct PhiloxPair {
    uint32_t a[4], b[4];
    uint32_t k0, k1;
    __device__ __forceinline__ void init(const PhiloxKey &key, uint64_t block) {
        a[0] = (uint32_t)block; a[1] = (uint32_t)(block >> 32); a[2] = key.site; a[3] = key.call;
        const uint64_t nb = block + 1;
        b[0] = (uint32_t)nb; b[1] = (uint32_t)(nb >> 32); b[2] = key.site; b[3] = key.call;
        k0 = key.k0; k1 = key.k1;
    }
    static __device__ __forceinline__ void round1(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
        constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
        const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
        const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    }
    __device__ __forceinline__ void fill() {
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            round1(a, k0, k1);
            round1(b, k0, k1);
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
    }
    __device__ __forceinline__ const uint32_t *words(int h) const { return h ? b : a; }
};

// the same for ONE block (a lane's 4 consecutive outputs: geometries with MI = 1)
struct PhiloxOne {
    uint32_t a[4];
    uint32_t k0, k1;
    __device__ __forceinline__ void init(const PhiloxKey &key, uint64_t block) {
        a[0] = (uint32_t)block; a[1] = (uint32_t)(block >> 32); a[2] = key.site; a[3] = key.call;
        k0 = key.k0; k1 = key.k1;
    }
    __device__ __forceinline__ void fill() {
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            PhiloxPair::round1(a, k0, k1);
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
    }
    __device__ __forceinline__ const uint32_t *words(int) const { return a; }
};

// TF Uint32ToFloat: 23 mantissa bits, [1,2) - 1
__device__ __forceinline__ float u32_to_uniform(uint32_t x) {
    return __uint_as_float(0x3f800000u | (x & 0x007fffffu)) - 1.0f;
}

// one word for flat element `idx` (slow generic path: recomputes the block)
__device__ __forceinline__ float philox_uniform_at(const PhiloxKey &key, uint64_t idx) {
    uint32_t w[4];
    philox_block(key, idx >> 2, w);
    return u32_to_uniform(w[idx & 3]);
}

// ln(u) for u in [2^-24, 1) and (sin, cos)(2 pi u) for u in [0, 1), specified operation by operation like the
// sigmoid (bm_numerics.h): correctly rounded fp32 mul / add / fma / div and integer bit operations only, so that
// oracle/bm_oracle.c (`pin_log_unit`, `pin_sincos_2pi`) reproduces every Normal draw BIT FOR BIT (round 3 used the
// device's logf / sincosf against glibc's: the Gaussian visible samples agreed to 1e-5 only and could flip a hidden
// bit downstream).  Absolute error < 1.2e-7 (log: relative), i.e. what a float32 libm gives.
//   log:    u = m 2^e with m in [1/sqrt2, sqrt2];  s = (m - 1) / (m + 1);  ln m = 2 s (1 + s^2/3 + ... + s^10/11);
//           ln u = e ln2_hi + (e ln2_lo + ln m)
//   sincos: t = 4 u, quadrant q = floor(t), f = t - q (exact), folded to [-1/2, 1/2]; x = f pi/2 in [-pi/4, pi/4];
//           Taylor to x^9 (sin) / x^10 (cos) in Horner form; the quadrant permutes / negates
__device__ __forceinline__ float pin_log_unit(float u) {
    const uint32_t b = __float_as_uint(u);
    int e = (int)(b >> 23) - 127;
    float m = __uint_as_float((b & 0x007fffffu) | 0x3f800000u);          // [1, 2)
    if (m > 1.41421356237309504880f) { m = m * 0.5f; e = e + 1; }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float t = s * s;
    float p = 1.0f / 11.0f;
    p = fmaf(p, t, 1.0f / 9.0f);
    p = fmaf(p, t, 1.0f / 7.0f);
    p = fmaf(p, t, 1.0f / 5.0f);
    p = fmaf(p, t, 1.0f / 3.0f);
    p = fmaf(p, t, 1.0f);
    const float lm = (2.0f * s) * p;
    const float ef = (float)e;
    return fmaf(ef, 0.693145751953125f, fmaf(ef, 1.42860682030941723212e-6f, lm));
}
__device__ __forceinline__ void pin_sincos_2pi(float u, float &sn, float &cs) {
    const float t = u * 4.0f;
    int q = (int)t;                                   // t >= 0: floor
    float f = t - (float)q;
    if (f > 0.5f) { f = f - 1.0f; q = q + 1; }
    const float x = f * 1.57079632679489661923f;
    const float x2 = x * x;
    float ps = 2.75573192239858906526e-6f;            // 1/9!
    ps = fmaf(ps, x2, -1.98412698412698412698e-4f);   // -1/7!
    ps = fmaf(ps, x2, 8.33333333333333333333e-3f);    // 1/5!
    ps = fmaf(ps, x2, -1.66666666666666666667e-1f);   // -1/3!
    const float sx = fmaf(x * x2, ps, x);
    float pc = -2.75573192239858906526e-7f;           // -1/10!
    pc = fmaf(pc, x2, 2.48015873015873015873e-5f);    // 1/8!
    pc = fmaf(pc, x2, -1.38888888888888888889e-3f);   // -1/6!
    pc = fmaf(pc, x2, 4.16666666666666666667e-2f);    // 1/4!
    pc = fmaf(pc, x2, -0.5f);
    const float cx = fmaf(pc, x2, 1.0f);
    switch (q & 3) {
        case 0: sn = sx; cs = cx; break;
        case 1: sn = cx; cs = -sx; break;
        case 2: sn = -sx; cs = -cx; break;
        default: sn = -cx; cs = sx; break;
    }
}

// TF BoxMullerFloat on a word pair -> two standard normals (sin first, cos second):
// u1 = max(U(x0), 1e-7), r = sqrt(-2 ln u1), (sin, cos)(2 pi U(x1)) r
__device__ __forceinline__ void box_muller(uint32_t x0, uint32_t x1, float &n0, float &n1) {
    const float eps = 1.0e-7f;
    float u1 = u32_to_uniform(x0);
    if (u1 < eps) u1 = eps;
    const float r = sqrtf(-2.0f * pin_log_unit(u1));
    float s, c;
    pin_sincos_2pi(u32_to_uniform(x1), s, c);
    n0 = s * r;
    n1 = c * r;
}

}  // namespace bm
