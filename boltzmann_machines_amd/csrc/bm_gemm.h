// bm_gemm.h — the fp32-MFMA tile engine every hot kernel of the engine is built on.
//
// All contractions of the reference hot path (tf.matmul at base_rbm.py:329-337,
// :447-448; dbm.py:390-425, :553-570, :650-694) are small dense fp32 GEMMs whose
// result feeds a nonlinearity + a Bernoulli draw or a parameter update.  They are
// computed with v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain) so that
// the result of every dot product is BIT-IDENTICAL to the sequential chain
//     acc = 0; for k in 0..K-1: acc = fmaf(p[k], q[k], acc)
// which is the order the CPU oracle uses ("canonical order").
//
// Geometry (wave64, gfx950):
//   output tile per workgroup : TI=64 (i, the contiguous output dim) x TJ=32 (j)
//   4 waves = 2 (i) x 2 (j); a wave owns 32 i x 16 j = two 16x16 MFMA tiles.
//   MFMA roles: A-operand <- P[i][k], B-operand <- Q[j][k].  The two MFMA tiles of
//   a wave interleave along i (tile t holds i = base + 2m + t, m = MFMA row), so
//     * ONE ds_read_b64 feeds the A operand of both MFMAs of a k-step, and
//     * accumulator lane (l&15)=j, group g=l>>4 holds the 8 CONSECUTIVE outputs
//       i = base + 8g + 2r + t  (r = register, t = tile) of output row j:
//       two 16-byte stores and exactly two Philox blocks per lane.
//   K is streamed in BK=32 chunks through double-buffered LDS; global->VGPR
//   prefetch runs PF chunks ahead (the whole problem is L2/MALL resident, the
//   loop is latency- not bandwidth-limited, one workgroup per CU).
//
// Operand storage:
//   P is always k-major  [k][i] (i contiguous): W for prop-up, the maintained
//     transpose Wt for prop-down, the hidden means for the outer products.
//   Q is k-major [k][j] (outer products: X, v) or x-major [j][k] (propagations:
//     rows of X / h, k contiguous).
// LDS strides make every fragment read conflict free:
//   P  (ds_read_b64, 64 banks): stride 96 == 32 (mod 64): lanes 0-15 cover 32
//      consecutive dwords of row k, lanes 16-31 the other 32 banks with row k+1.
//   Q KM (ds_read_b32, 32 banks): stride 48 == 16 (mod 32).
//   Q XM (ds_read_b32): stride 34 == 2 (mod 4): bank = (2j + k) mod 32 is a
//      bijection on 16 j x 2 k.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bm {

// VALU side work threaded through the main loop, one call per K step (default: none)
struct NoSide { __device__ __forceinline__ void step() {} };

// compile-time ablation mask (template parameter ABL, 0 in the product; tools/probe_act.hip
// instantiates other values to price each pipeline stage)
#define BM_ABL(bit) ((ABL >> (bit)) & 1)

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TI = 64;    // tile extent along i
constexpr int BK = 64;    // K chunk
// tile extent along j = 32 * NJ (NJ = 16-wide j sub-tiles per wave): NJ = 1 for the
// propagations (batch 512 -> 16 x 16 = 256 tiles = one per CU), NJ = 2 for the outer
// products (64 x 64 tiles: 4 MFMAs per fragment pair, one wave of 208 tiles at 784x1024)
constexpr int TJ = 32;    // NJ = 1 extent (kept for the host-side grid helpers)
constexpr int NT = 256;   // threads per workgroup

enum : int { KM = 0, XM = 1 };

constexpr int P_STRIDE    = TI + 32;   // 96
constexpr int Q_STRIDE_XM = BK + 2;    // 66
constexpr int P_BUF = BK * P_STRIDE;
constexpr int NBUF = 3;   // LDS ring depth
template <int NJ> struct TileGeom {
    static constexpr int TJn = 32 * NJ;
    static constexpr int Q_STRIDE_KM = TJn + 16;   // == 16 (mod 32)
    static constexpr int Q_BUF = (BK * Q_STRIDE_KM > TJn * Q_STRIDE_XM) ? BK * Q_STRIDE_KM : TJn * Q_STRIDE_XM;
    static constexpr int SMEM_FLOATS = NBUF * (P_BUF + Q_BUF);
};
constexpr int SMEM_FLOATS = TileGeom<1>::SMEM_FLOATS;      // 108 KiB
constexpr int SMEM_FLOATS2 = TileGeom<2>::SMEM_FLOATS;     // 132 KiB

struct Operand {
    const float *ptr;
    int ld;    // leading dimension (floats)
    int nx;    // extent along x (i for P, j for Q)
    int vec;   // 1: 16-byte loads are legal (ptr 16B aligned, ld % 4 == 0)
};

static inline Operand make_operand(const float *p, int ld, int nx) {
    Operand o;
    o.ptr = p; o.ld = ld; o.nx = nx;
    o.vec = (((uintptr_t)p & 15u) == 0 && (ld & 3) == 0) ? 1 : 0;
    return o;
}

// 16 zero bytes for branch-free guarded scalar loads (colsum_kernel)
static __device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// host: can this operand take the branch-free load path?
static inline bool operand_fast(const Operand &o, int layout, int K) {
    if (!o.ptr) return true;                      // absent segment
    if (!o.vec) return false;
    return layout == KM ? (o.nx % 4 == 0) : (K % 4 == 0);
}

__device__ __forceinline__ float4 load4_guard(const float *p, bool row_ok, int col, int ncols, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row_ok) {
        if (vec && col + 3 < ncols) {
            v = *reinterpret_cast<const float4 *>(p);
        } else {
            if (col     < ncols) v.x = p[0];
            if (col + 1 < ncols) v.y = p[1];
            if (col + 2 < ncols) v.z = p[2];
            if (col + 3 < ncols) v.w = p[3];
        }
    }
    return v;
}

// global -> registers for one BK chunk of one operand tile (TX = tile extent along x).
// FAST: every float4 is either fully inside or fully outside the operand (host
// guarantees 16B alignment, ld % 4 == 0 and a contiguous extent % 4 == 0), so the
// load is unconditional and branch free: indices are clamped into range; the K tail is
// zeroed when the set is stored to LDS, and x-tail garbage only reaches outputs
// i >= I / j >= J, which are never stored.
template <int L, int TX, bool FAST>
__device__ __forceinline__ void g2r(float4 (&reg)[TX * BK / (4 * NT)], const float *ptr, int ld, int nx, int vec,
                                    int x0, int k0, int K, int tid) {
    constexpr int NV = TX * BK / (4 * NT);   // float4 per thread
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int f = tid + n * NT;
        int k, x;
        if (L == KM) {
            const int row = f / (TX / 4), c4 = f % (TX / 4);
            k = k0 + row; x = x0 + c4 * 4;
        } else {
            const int row = f / (BK / 4), c4 = f % (BK / 4);
            x = x0 + row; k = k0 + c4 * 4;
        }
        if (FAST) {
            const int kc = (L == KM) ? min(k, K - 1) : min(k, K - 4);
            const int xc = (L == KM) ? min(x, nx - 4) : min(x, nx - 1);
            const float *pc = (L == KM) ? ptr + (size_t)kc * ld + xc : ptr + (size_t)xc * ld + kc;
            reg[n] = *reinterpret_cast<const float4 *>(pc);
        } else if (L == KM) {
            reg[n] = load4_guard(ptr + (size_t)k * ld + x, k < K, x, nx, vec);
        } else {
            reg[n] = load4_guard(ptr + (size_t)x * ld + k, x < nx, k, K, vec);
        }
    }
}

// registers -> LDS.  kz = K - k0 (rows/cols of this chunk at k >= K are zeroed; only the
// FAST path needs it, the guarded loads already returned zeros).
template <int L, int TX, int STRIDE_K, bool FAST, bool SIGN = false>
__device__ __forceinline__ void r2s(const float4 (&reg)[TX * BK / (4 * NT)], float *s, int tid, int kz,
                                    uint32_t signmask = 0u) {
    constexpr int NV = TX * BK / (4 * NT);
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int f = tid + n * NT;
        float4 v = reg[n];
        if (SIGN) {              // branch-free (a branch would split the step's scheduling region)
            v.x = __uint_as_float(__float_as_uint(v.x) ^ signmask);
            v.y = __uint_as_float(__float_as_uint(v.y) ^ signmask);
            v.z = __uint_as_float(__float_as_uint(v.z) ^ signmask);
            v.w = __uint_as_float(__float_as_uint(v.w) ^ signmask);
        }
        if (L == KM) {
            const int row = f / (TX / 4), c4 = f % (TX / 4);
            if (FAST && row >= kz) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(s + row * STRIDE_K + c4 * 4) = v;
        } else {
            const int row = f / (BK / 4), c4 = f % (BK / 4);
            if (FAST && c4 * 4 >= kz) v = make_float4(0.f, 0.f, 0.f, 0.f);
            float2 *d = reinterpret_cast<float2 *>(s + row * Q_STRIDE_XM + c4 * 4);
            d[0] = make_float2(v.x, v.y);
            d[1] = make_float2(v.z, v.w);
        }
    }
}

// MFMA operand fragments of one BK chunk for this wave (48 VGPRs at BK = 64, NJ = 1)
template <int NJ> struct Frags {
    float2 p[BK / 4];       // p[kk] = P[k = 4kk+g][i = base + 2*l15 + {0,1}]
    float q[BK / 4][NJ];    // q[kk][n] = Q[j = 16n + l15][k = 4kk+g]
};

template <int QL, int NJ, int ABL = 0>
__device__ __forceinline__ void read_frags(Frags<NJ> &f, const float *sP, const float *sQ, int wi, int wj, int lane) {
    using G = TileGeom<NJ>;
    const int g = lane >> 4, l15 = lane & 15;
    const float *pP = sP + g * P_STRIDE + wi * 32 + 2 * l15;
    const float *pQ = (QL == KM) ? sQ + g * G::Q_STRIDE_KM + wj * 16 * NJ + l15
                                 : sQ + (wj * 16 * NJ + l15) * Q_STRIDE_XM + g;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
        if (BM_ABL(3)) { f.p[kk] = make_float2(1.f, 2.f); f.q[kk][0] = 1.f; continue; }
        // (hipcc fuses pairs of these into ds_read2st64_b64; keeping them as separate ds_read_b64
        // was measured 34 % SLOWER in the loop, so the fused form stays)
        f.p[kk] = *reinterpret_cast<const float2 *>(pP + kk * 4 * P_STRIDE);
#pragma unroll
        for (int n = 0; n < NJ; ++n)
            f.q[kk][n] = (QL == KM) ? pQ[kk * 4 * G::Q_STRIDE_KM + 16 * n] : pQ[16 * n * Q_STRIDE_XM + kk * 4];
    }
}

// acc[t][n] += P-frag(t) x Q-frag(n)
template <int NJ, int ABL = 0>
__device__ __forceinline__ void mfma_frags(f32x4 (&acc)[2][NJ], const Frags<NJ> &f) {
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
#pragma unroll
        for (int n = 0; n < NJ; ++n) {
            const float q = f.q[kk][n];
            if (BM_ABL(1)) { acc[0][n][0] += f.p[kk].x * q; acc[1][n][0] += f.p[kk].y; continue; }
            acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.p[kk].x, q, acc[0][n], 0, 0, 0);
            acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.p[kk].y, q, acc[1][n], 0, 0, 0);
        }
    }
}

// one register set = one BK chunk of both operand tiles, global -> VGPR staging
template <int NJ> struct ChunkRegs {
    float4 p[TI * BK / (4 * NT)];
    float4 q[32 * NJ * BK / (4 * NT)];
};

// The K range of a contraction: segment 1 followed by an optional segment 2 with its own
// operands (DBM two-sided layer input; positive then negative phase of the outer
// products), streamed as ONE continuous pipeline (one fill, one drain).
struct KRange {
    Operand P1, Q1; int K1;
    Operand P2, Q2; int K2;     // K2 == 0: absent
    float sgn2;                 // +1 or -1: sign applied to segment-2 products
};

// branch-free wave-uniform selects (a branch inside a step would split its scheduling region)
__device__ __forceinline__ int sel_i(int a, int b, int m) { return a ^ ((a ^ b) & m); }
__device__ __forceinline__ const float *sel_p(const float *a, const float *b, int m) {
    const uintptr_t ua = (uintptr_t)a, ub = (uintptr_t)b;
    return (const float *)(ua ^ ((ua ^ ub) & (uintptr_t)(intptr_t)m));
}

template <int QL, int NJ, bool FAST, bool SEG2>
__device__ __forceinline__ void load_chunk(ChunkRegs<NJ> &r, const KRange &kr, int nch1, int i0, int j0, int c, int tid) {
    if (!SEG2) {
        g2r<KM, TI, FAST>(r.p, kr.P1.ptr, kr.P1.ld, kr.P1.nx, kr.P1.vec, i0, c * BK, kr.K1, tid);
        g2r<QL, 32 * NJ, FAST>(r.q, kr.Q1.ptr, kr.Q1.ld, kr.Q1.nx, kr.Q1.vec, j0, c * BK, kr.K1, tid);
    } else {
        const int m = -(int)((c >= nch1) & (kr.K2 > 0));   // all-ones in segment 2 (wave-uniform)
        const int kc = c - (nch1 & m);
        const int K = sel_i(kr.K1, kr.K2, m);
        g2r<KM, TI, FAST>(r.p, sel_p(kr.P1.ptr, kr.P2.ptr, m), sel_i(kr.P1.ld, kr.P2.ld, m),
                          sel_i(kr.P1.nx, kr.P2.nx, m), sel_i(kr.P1.vec, kr.P2.vec, m), i0, kc * BK, K, tid);
        g2r<QL, 32 * NJ, FAST>(r.q, sel_p(kr.Q1.ptr, kr.Q2.ptr, m), sel_i(kr.Q1.ld, kr.Q2.ld, m),
                               sel_i(kr.Q1.nx, kr.Q2.nx, m), sel_i(kr.Q1.vec, kr.Q2.vec, m), j0, kc * BK, K, tid);
    }
}

template <int QL, int NJ, bool FAST, bool SEG2>
__device__ __forceinline__ void store_chunk(const ChunkRegs<NJ> &r, const KRange &kr, int nch1, int c,
                                            float *sP, float *sQ, int tid) {
    const int m = SEG2 ? -(int)((c >= nch1) & (kr.K2 > 0)) : 0;
    const int kz = sel_i(kr.K1, kr.K2, m) - (c - (nch1 & m)) * BK;
    r2s<KM, TI, P_STRIDE, FAST>(r.p, sP, tid, kz);
    // the sign of segment 2 (-1 for the negative CD phase) is applied once per element here,
    // off the MFMA dependency chain: fma(p, -q, acc) == acc - p*q exactly
    const uint32_t sm = (SEG2 && kr.sgn2 < 0.f) ? (0x80000000u & (uint32_t)m) : 0u;
    r2s<QL, 32 * NJ, TileGeom<NJ>::Q_STRIDE_KM, FAST, SEG2>(r.q, sQ, tid, kz, sm);
}

// acc += sum_k P[k][i] * Q[j][k] over the K range, k ascending (canonical order).
//
// One wave per SIMD has nobody to hide behind, so the loop is software-pipelined by
// hand over a 3-deep LDS ring.  In step c (between barriers B(c-1) and B(c)) a wave
//   * runs the MFMAs of chunk c on fragments F(c) that are ALREADY in registers,
//   * reads the fragments F(c+1) from LDS slot (c+1)%3 (published by B(c-1)),
//   * stores chunk c+2 (global data that arrived in registers) to LDS slot (c+2)%3
//     (last read for F(c-1), complete before B(c-2)),
//   * re-issues the global loads of chunk c+4 into the register set just stored,
//   * runs one call of the VALU side work (Philox round).
// sched_group_barrier pins that issue order: hipcc otherwise issues the LDS traffic
// AFTER the MFMAs and the two phases run back to back.  No load sits under a branch
// (chunks past K are clamped / zero-filled), so all waits are counted.
template <int QL, int NJ, bool FAST, bool SEG2, int ABL = 0, class Side = NoSide>
__device__ __forceinline__ void mainloop(f32x4 (&acc)[2][NJ], const KRange &kr, int i0, int j0, float *smem,
                                         Side &side, long long *stamps = nullptr) {
#ifdef BM_PROBE
#define BM_MSTAMP(n) do { if (stamps && threadIdx.x == 0) stamps[n] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define BM_MSTAMP(n) do {} while (0)
#endif
    using G = TileGeom<NJ>;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w & 1, wj = w >> 1;
    float *sP = smem, *sQ = smem + NBUF * P_BUF;
    const int nch1 = (kr.K1 + BK - 1) / BK;
    const int nch = nch1 + (SEG2 ? (kr.K2 + BK - 1) / BK : 0);
    ChunkRegs<NJ> g0, g1;      // named sets (never arrays: must stay in VGPRs)
    Frags<NJ> fa, fb;
    BM_MSTAMP(0);
    {   // pipeline fill: the four chunk loads go out back to back (ONE memory round trip);
        // chunks 0/1 pass through two prologue-only sets, chunks 2/3 land in the loop's sets
        ChunkRegs<NJ> ga, gb;
        load_chunk<QL, NJ, FAST, SEG2>(ga, kr, nch1, i0, j0, 0, tid);
        load_chunk<QL, NJ, FAST, SEG2>(gb, kr, nch1, i0, j0, 1, tid);
        load_chunk<QL, NJ, FAST, SEG2>(g0, kr, nch1, i0, j0, 2, tid);
        load_chunk<QL, NJ, FAST, SEG2>(g1, kr, nch1, i0, j0, 3, tid);
        store_chunk<QL, NJ, FAST, SEG2>(ga, kr, nch1, 0, sP, sQ, tid);
        store_chunk<QL, NJ, FAST, SEG2>(gb, kr, nch1, 1, sP + P_BUF, sQ + G::Q_BUF, tid);
    }
    BM_MSTAMP(1);
    __syncthreads();
    read_frags<QL, NJ, ABL>(fa, sP, sQ, wi, wj, lane);
    BM_MSTAMP(2);
    int cc = 0, b1 = 1, b2 = 2;   // LDS slots of chunk cc+1 / cc+2
    // masks: 0x008 MFMA, 0x100 DS read, 0x200 DS write, 0x020 VMEM read
#define BM_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0);
#ifndef BM_SCHED_VARIANT
#define BM_SCHED_VARIANT 0
#endif
#if BM_SCHED_VARIANT == 0
#define BM_SCHED_STEP                                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) { BM_SG(0x008, NJ) BM_SG(0x100, 1 + NJ) }    \
    _Pragma("unroll") for (int s_ = 0; s_ < 4 + 2 * NJ; ++s_) { BM_SG(0x008, 2) BM_SG(0x200, 1) } \
    _Pragma("unroll") for (int s_ = 0; s_ < 4 + 2 * NJ; ++s_) { BM_SG(0x008, 1) BM_SG(0x020, 1) } \
    BM_SG(0x008, 18 * NJ - 12)
#elif BM_SCHED_VARIANT == 1   /* stores first */
#define BM_SCHED_STEP                                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < 4 + 2 * NJ; ++s_) { BM_SG(0x008, 1) BM_SG(0x200, 1) } \
    _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) { BM_SG(0x008, 2 * NJ) BM_SG(0x100, 1 + NJ) } \
    _Pragma("unroll") for (int s_ = 0; s_ < 4 + 2 * NJ; ++s_) { BM_SG(0x008, 1) BM_SG(0x020, 1) } \
    BM_SG(0x008, 12 * NJ - 8)
#elif BM_SCHED_VARIANT == 2   /* global loads first */
#define BM_SCHED_STEP                                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < 4 + 2 * NJ; ++s_) { BM_SG(0x008, 1) BM_SG(0x020, 1) } \
    _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) { BM_SG(0x008, NJ) BM_SG(0x100, 1 + NJ) }    \
    _Pragma("unroll") for (int s_ = 0; s_ < 4 + 2 * NJ; ++s_) { BM_SG(0x008, 2) BM_SG(0x200, 1) } \
    BM_SG(0x008, 18 * NJ - 12)
#elif BM_SCHED_VARIANT == 3   /* reads spread 1:1 over the MFMAs */
#define BM_SCHED_STEP                                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < 8 * (1 + NJ); ++s_) { BM_SG(0x008, 1) BM_SG(0x100, 1) } \
    _Pragma("unroll") for (int s_ = 0; s_ < 4 + 2 * NJ; ++s_) { BM_SG(0x008, 1) BM_SG(0x200, 1) } \
    _Pragma("unroll") for (int s_ = 0; s_ < 4 + 2 * NJ; ++s_) { BM_SG(0x008, 1) BM_SG(0x020, 1) } \
    BM_SG(0x008, 32 * NJ)
#elif BM_SCHED_VARIANT == 4   /* stores and loads paired, after the reads */
#define BM_SCHED_STEP                                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) { BM_SG(0x008, NJ) BM_SG(0x100, 1 + NJ) }    \
    _Pragma("unroll") for (int s_ = 0; s_ < 4 + 2 * NJ; ++s_) { BM_SG(0x008, 2) BM_SG(0x200, 1) BM_SG(0x020, 1) } \
    BM_SG(0x008, 32 * NJ)
#endif
#define BM_STEP(FC, FN, G_)                                                                       \
    {                                                                                             \
        if (!BM_ABL(3)) read_frags<QL, NJ, ABL>(FN, sP + b1 * P_BUF, sQ + b1 * G::Q_BUF, wi, wj, lane); \
        if (!BM_ABL(2)) store_chunk<QL, NJ, FAST, SEG2>(G_, kr, nch1, cc + 2, sP + b2 * P_BUF, sQ + b2 * G::Q_BUF, tid); \
        if (!BM_ABL(0)) load_chunk<QL, NJ, FAST, SEG2>(G_, kr, nch1, i0, j0, cc + 4, tid);        \
        side.step();                                                                              \
        mfma_frags<NJ, ABL>(acc, FC);                                                             \
        BM_SCHED_STEP                                                                             \
        if (!BM_ABL(5)) __syncthreads();                                                          \
        b1 = b2;                                                                                  \
        b2 = (b2 == NBUF - 1) ? 0 : b2 + 1;                                                       \
        ++cc;                                                                                     \
    }
    // steady state: steps 0 .. nch-3 carry the full LDS / global traffic; the last two steps
    // (pipeline drain) only have fragments to read and MFMAs to issue
    const int full = (nch > 2) ? nch - 2 : 0;
    for (int pi = 0; pi < full / 2; ++pi) {
        BM_STEP(fa, fb, g0)
        BM_STEP(fb, fa, g1)
    }
    if (full & 1) {
        BM_STEP(fa, fb, g0)
        fa = fb;               // keep the current fragments in `fa` for the drain (once per kernel)
    }
    BM_MSTAMP(3);
    if (nch >= 2) {
        if (!BM_ABL(3)) read_frags<QL, NJ, ABL>(fb, sP + b1 * P_BUF, sQ + b1 * G::Q_BUF, wi, wj, lane);
        side.step();
        mfma_frags<NJ, ABL>(acc, fa);
        side.step();
        mfma_frags<NJ, ABL>(acc, fb);
    } else {
        side.step();
        mfma_frags<NJ, ABL>(acc, fa);
    }
    __syncthreads();           // the LDS ring may be refilled by a following pipeline
    BM_MSTAMP(4);
#undef BM_STEP
#undef BM_SCHED_STEP
#undef BM_SG
}

// the 8 consecutive outputs of a lane for j sub-tile n: v[e], e = 2r + t  <->  i = ib + e
template <int NJ>
__device__ __forceinline__ void lane_outputs(const f32x4 (&acc)[2][NJ], int n, float (&v)[8]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[2 * r] = acc[0][n][r];
        v[2 * r + 1] = acc[1][n][r];
    }
}

// XCD-aware block -> tile map: blocks are dispatched round-robin over the 8 XCDs
// (block b -> XCD b % 8, MI355X_MICROARCH.md), so consecutive logical tiles
// t (which share the P panel = same i-tile) are placed on ONE XCD's L2.
__device__ __forceinline__ void block_to_tile(int tiles_j, int &ti, int &tj, int skip = 0, int trail = 0) {
    // `skip` leading / `trail` trailing non-tile workgroups in the launch
    const int nb = gridDim.x - skip - trail, b = blockIdx.x - skip;
    const int q = nb / 8, r = nb % 8, xcd = b % 8;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    ti = t / tiles_j;
    tj = t % tiles_j;
}

}  // namespace bm
