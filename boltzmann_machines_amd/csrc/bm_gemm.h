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
constexpr int TJ = 32;    // tile extent along j
constexpr int BK = 64;    // K chunk
constexpr int NT = 256;   // threads per workgroup

enum : int { KM = 0, XM = 1 };

constexpr int P_STRIDE    = TI + 32;   // 96
constexpr int Q_STRIDE_KM = TJ + 16;   // 48
constexpr int Q_STRIDE_XM = BK + 2;    // 34
constexpr int P_BUF = BK * P_STRIDE;                                                              // 3072
constexpr int Q_BUF = (BK * Q_STRIDE_KM > TJ * Q_STRIDE_XM) ? BK * Q_STRIDE_KM : TJ * Q_STRIDE_XM;  // 1536
constexpr int NBUF = 3;   // LDS ring depth
constexpr int SMEM_FLOATS = NBUF * (P_BUF + Q_BUF);                                               // 108 KiB at BK = 64

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

// global -> registers for one BK chunk of one operand tile (TX = TI or TJ).
// FAST: every float4 is either fully inside or fully outside the operand (host
// guarantees 16B alignment, ld % 4 == 0 and a contiguous extent % 4 == 0), so the
// load is unconditional and branch free: an out-of-range lane reads g_zero16.
// That keeps PF chunks of loads in flight with counted vmcnt waits only.
template <int L, int TX, bool FAST>
__device__ __forceinline__ void g2r(float4 (&reg)[TX * BK / (4 * NT)], const Operand &op, int x0, int k0, int K, int tid) {
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
        const float *p = (L == KM) ? op.ptr + (size_t)k * op.ld + x : op.ptr + (size_t)x * op.ld + k;
        if (FAST) {
            // clamp into range (always a legal 16-byte load, no branch, no select on the
            // pointer); the K tail is zeroed when the set is stored to LDS (r2s), and
            // x-tail garbage only reaches outputs i >= I / j >= J, which are never stored.
            const int kc = (L == KM) ? min(k, K - 1) : min(k, K - 4);
            const int xc = (L == KM) ? min(x, op.nx - 4) : min(x, op.nx - 1);
            const float *pc = (L == KM) ? op.ptr + (size_t)kc * op.ld + xc : op.ptr + (size_t)xc * op.ld + kc;
            reg[n] = *reinterpret_cast<const float4 *>(pc);
        } else if (L == KM) {
            reg[n] = load4_guard(p, k < K, x, op.nx, op.vec);
        } else {
            reg[n] = load4_guard(p, x < op.nx, k, K, op.vec);
        }
    }
}

// registers -> LDS.  kz = K - k0 (rows/cols of this chunk at k >= K are zeroed; only the
// FAST path needs it, the guarded loads already returned zeros).
template <int L, int TX, bool FAST>
__device__ __forceinline__ void r2s(const float4 (&reg)[TX * BK / (4 * NT)], float *s, int tid, int kz) {
    constexpr int NV = TX * BK / (4 * NT);
    constexpr int STRIDE_K = (TX == TI) ? P_STRIDE : Q_STRIDE_KM;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int f = tid + n * NT;
        float4 v = reg[n];
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

// MFMA operand fragments of one BK chunk for this wave (48 VGPRs at BK = 64)
struct Frags {
    float2 p[BK / 4];   // p[kk] = P[k = 4kk+g][i = base + 2*l15 + {0,1}]
    float q[BK / 4];    // q[kk] = Q[j = l15][k = 4kk+g]
};

template <int QL, int ABL = 0>
__device__ __forceinline__ void read_frags(Frags &f, const float *sP, const float *sQ, int wi, int wj, int lane) {
    const int g = lane >> 4, l15 = lane & 15;
    const float *pP = sP + g * P_STRIDE + wi * 32 + 2 * l15;
    const float *pQ = (QL == KM) ? sQ + g * Q_STRIDE_KM + wj * 16 + l15
                                 : sQ + (wj * 16 + l15) * Q_STRIDE_XM + g;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
        if (BM_ABL(3)) { f.p[kk] = make_float2(1.f, 2.f); f.q[kk] = 1.f; continue; }
        f.p[kk] = *reinterpret_cast<const float2 *>(pP + kk * 4 * P_STRIDE);
        f.q[kk] = (QL == KM) ? pQ[kk * 4 * Q_STRIDE_KM] : pQ[kk * 4];
    }
}

template <int ABL = 0>
__device__ __forceinline__ void mfma_frags(f32x4 (&acc)[2], const Frags &f) {
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
        if (BM_ABL(1)) { acc[0][0] += f.p[kk].x * f.q[kk]; acc[1][0] += f.p[kk].y; continue; }
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.p[kk].x, f.q[kk], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.p[kk].y, f.q[kk], acc[1], 0, 0, 0);
    }
}

// one register set = one BK chunk of both operand tiles, global -> VGPR staging
struct ChunkRegs {
    float4 p[TI * BK / (4 * NT)];
    float4 q[TJ * BK / (4 * NT)];
};

template <int QL, bool FAST>
__device__ __forceinline__ void load_chunk(ChunkRegs &r, const Operand &P, const Operand &Q, int K,
                                           int i0, int j0, int c, int tid) {
    g2r<KM, TI, FAST>(r.p, P, i0, c * BK, K, tid);
    g2r<QL, TJ, FAST>(r.q, Q, j0, c * BK, K, tid);
}

template <int QL, bool FAST>
__device__ __forceinline__ void store_chunk(const ChunkRegs &r, float *sP, float *sQ, int tid, int kz) {
    r2s<KM, TI, FAST>(r.p, sP, tid, kz);
    r2s<QL, TJ, FAST>(r.q, sQ, tid, kz);
}

// acc += sum_k P[k][i] * Q[j][k] over k in [0, K), k ascending (canonical order).
//
// One wave per SIMD has nobody to hide behind, so the loop is software-pipelined by
// hand over a 3-deep LDS ring.  In step c (between barriers B(c-1) and B(c)) a wave
//   * runs the 32 MFMAs of chunk c on fragments F(c) that are ALREADY in registers,
//   * reads the fragments F(c+1) from LDS slot (c+1)%3 (published by B(c-1)),
//   * stores chunk c+2 (global data that arrived in registers) to LDS slot (c+2)%3
//     (last read for F(c-1), complete before B(c-2)),
//   * re-issues the global loads of chunk c+4 into the register set just stored.
// LDS traffic, global loads and the barrier all overlap the MFMA stream of the same
// wave; the only exposed latency is the barrier skew.  No load sits under a branch
// (chunks past K are clamped / zero-filled), so all waits are counted.
template <int QL, bool FAST, int ABL = 0, class Side = NoSide>
__device__ __forceinline__ void mainloop(f32x4 (&acc)[2], const Operand &P, const Operand &Q, int K,
                                         int i0, int j0, float *smem, Side &side, long long *stamps = nullptr) {
#ifdef BM_PROBE
#define BM_MSTAMP(n) do { if (stamps && threadIdx.x == 0) stamps[n] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define BM_MSTAMP(n) do {} while (0)
#endif
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w & 1, wj = w >> 1;
    float *sP = smem, *sQ = smem + NBUF * P_BUF;
    const int nch = (K + BK - 1) / BK;
    ChunkRegs g0, g1;          // named sets (never arrays: must stay in VGPRs)
    Frags fa, fb;
    BM_MSTAMP(0);
    load_chunk<QL, FAST>(g0, P, Q, K, i0, j0, 0, tid);
    load_chunk<QL, FAST>(g1, P, Q, K, i0, j0, 1, tid);
    store_chunk<QL, FAST>(g0, sP, sQ, tid, K);
    load_chunk<QL, FAST>(g0, P, Q, K, i0, j0, 2, tid);
    store_chunk<QL, FAST>(g1, sP + P_BUF, sQ + Q_BUF, tid, K - BK);
    load_chunk<QL, FAST>(g1, P, Q, K, i0, j0, 3, tid);
    BM_MSTAMP(1);
    __syncthreads();
    read_frags<QL, ABL>(fa, sP, sQ, wi, wj, lane);
    BM_MSTAMP(2);
    int cc = 0, b1 = 1, b2 = 2;   // LDS slots of chunk cc+1 / cc+2
    // Desired issue order inside a step (hipcc otherwise issues the LDS traffic AFTER the
    // MFMAs and the two phases run back to back): LDS reads of F(c+1) right behind the
    // barrier, then the stores of chunk c+2, then the global re-loads, each group
    // threaded between MFMAs so the matrix pipe never waits for the LDS pipe.
    // masks: 0x008 MFMA, 0x100 DS read, 0x200 DS write, 0x020 VMEM read
#define BM_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0);
#define BM_SCHED_STEP                                                                        \
    _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) { BM_SG(0x008, 1) BM_SG(0x100, 2) }     \
    _Pragma("unroll") for (int s_ = 0; s_ < 6; ++s_) { BM_SG(0x008, 2) BM_SG(0x200, 1) }     \
    _Pragma("unroll") for (int s_ = 0; s_ < 6; ++s_) { BM_SG(0x008, 1) BM_SG(0x020, 1) }     \
    BM_SG(0x008, 6)
#define BM_STEP(FC, FN, G)                                                                   \
    {                                                                                        \
        if (!BM_ABL(3)) read_frags<QL, ABL>(FN, sP + b1 * P_BUF, sQ + b1 * Q_BUF, wi, wj, lane); \
        if (!BM_ABL(2)) store_chunk<QL, FAST>(G, sP + b2 * P_BUF, sQ + b2 * Q_BUF, tid, K - (cc + 2) * BK); \
        if (!BM_ABL(0)) load_chunk<QL, FAST>(G, P, Q, K, i0, j0, cc + 4, tid);               \
        side.step();                                                                         \
        mfma_frags<ABL>(acc, FC);                                                            \
        BM_SCHED_STEP                                                                        \
        if (!BM_ABL(5)) __syncthreads();                                                     \
        b1 = b2;                                                                             \
        b2 = (b2 == NBUF - 1) ? 0 : b2 + 1;                                                  \
        ++cc;                                                                                \
    }
    for (int pi = 0; pi < nch / 2; ++pi) {
        BM_STEP(fa, fb, g0)
        BM_STEP(fb, fa, g1)
    }
    BM_MSTAMP(3);
    if (nch & 1) BM_STEP(fa, fb, g0)
    BM_MSTAMP(4);
#undef BM_STEP
#undef BM_SCHED_STEP
#undef BM_SG
}

// the 8 consecutive outputs of a lane: v[e], e = 2r + t  <->  i = ib + e
__device__ __forceinline__ void lane_outputs(const f32x4 (&acc)[2], float (&v)[8]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[2 * r] = acc[0][r];
        v[2 * r + 1] = acc[1][r];
    }
}

// XCD-aware block -> tile map: blocks are dispatched round-robin over the 8 XCDs
// (block b -> XCD b % 8, MI355X_MICROARCH.md), so consecutive logical tiles
// t (which share the P panel = same i-tile) are placed on ONE XCD's L2.
__device__ __forceinline__ void block_to_tile(int tiles_j, int &ti, int &tj, int skip = 0) {
    const int nb = gridDim.x - skip, b = blockIdx.x - skip;   // `skip` leading non-tile workgroups
    const int q = nb / 8, r = nb % 8, xcd = b % 8;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    ti = t / tiles_j;
    tj = t % tiles_j;
}

}  // namespace bm
