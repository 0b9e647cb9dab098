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

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TI = 64;    // tile extent along i
constexpr int TJ = 32;    // tile extent along j
constexpr int BK = 32;    // K chunk
constexpr int NT = 256;   // threads per workgroup
constexpr int PF = 3;     // global->register prefetch distance (chunks)

enum : int { KM = 0, XM = 1 };

constexpr int P_STRIDE    = TI + 32;   // 96
constexpr int Q_STRIDE_KM = TJ + 16;   // 48
constexpr int Q_STRIDE_XM = BK + 2;    // 34
constexpr int P_BUF = BK * P_STRIDE;                                                              // 3072
constexpr int Q_BUF = (BK * Q_STRIDE_KM > TJ * Q_STRIDE_XM) ? BK * Q_STRIDE_KM : TJ * Q_STRIDE_XM;  // 1536
constexpr int SMEM_FLOATS = 2 * (P_BUF + Q_BUF);                                                  // 36 KiB

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
__device__ __forceinline__ void g2r(float4 (&reg)[TX / 32], const Operand &op, int x0, int k0, int K, int tid) {
    constexpr int NV = TX / 32;   // float4 per thread
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
__device__ __forceinline__ void r2s(const float4 (&reg)[TX / 32], float *s, int tid, int kz) {
    constexpr int NV = TX / 32;
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

// one BK chunk of MFMAs for this wave: acc[t] += P-frag(t) x Q-frag
template <int QL>
__device__ __forceinline__ void compute_chunk(f32x4 (&acc)[2], const float *sP, const float *sQ,
                                              int wi, int wj, int lane) {
    const int g = lane >> 4, l15 = lane & 15;
    const float *pP = sP + g * P_STRIDE + wi * 32 + 2 * l15;
    const float *pQ = (QL == KM) ? sQ + g * Q_STRIDE_KM + wj * 16 + l15
                                 : sQ + (wj * 16 + l15) * Q_STRIDE_XM + g;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
        const float q  = (QL == KM) ? pQ[kk * 4 * Q_STRIDE_KM] : pQ[kk * 4];
        const float2 p = *reinterpret_cast<const float2 *>(pP + kk * 4 * P_STRIDE);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p.x, q, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p.y, q, acc[1], 0, 0, 0);
    }
}

// acc += sum_k P[k][i] * Q[j][k] over k in [0, K), k ascending (canonical order).
// Software pipeline: register set (c % PF) holds chunk c; at step c the set is
// refilled with chunk c+PF (loads stay in flight for PF-1 steps), chunk c is
// consumed from LDS buffer c&1 and chunk c+1 moves registers -> LDS buffer (c+1)&1.
// No load sits under a branch (chunks past K read g_zero16 / the guarded path
// returns zeros), so the only waits are counted vmcnt for the set being stored.
// one register set = one BK chunk of both operand tiles (3 x 16 B per thread)
struct ChunkRegs {
    float4 p[TI / 32];
    float4 q[TJ / 32];
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

template <int QL, bool FAST>
__device__ __forceinline__ void mainloop(f32x4 (&acc)[2], const Operand &P, const Operand &Q, int K,
                                         int i0, int j0, float *smem) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w & 1, wj = w >> 1;
    float *sP = smem, *sQ = smem + 2 * P_BUF;
    const int nch = (K + BK - 1) / BK;
    static_assert(PF == 3, "the step sequence below is written for PF == 3");
    ChunkRegs r0, r1, r2;          // named sets (not an array): must stay in VGPRs, never scratch
    load_chunk<QL, FAST>(r0, P, Q, K, i0, j0, 0, tid);
    load_chunk<QL, FAST>(r1, P, Q, K, i0, j0, 1, tid);
    load_chunk<QL, FAST>(r2, P, Q, K, i0, j0, 2, tid);
    store_chunk<QL, FAST>(r0, sP, sQ, tid, K);
    __syncthreads();
    int cc = 0, cur = 0;
#define BM_STEP(RU, RN)                                                                     \
    {                                                                                       \
        load_chunk<QL, FAST>(RU, P, Q, K, i0, j0, cc + PF, tid);                            \
        compute_chunk<QL>(acc, sP + cur * P_BUF, sQ + cur * Q_BUF, wi, wj, lane);           \
        store_chunk<QL, FAST>(RN, sP + (cur ^ 1) * P_BUF, sQ + (cur ^ 1) * Q_BUF, tid,     \
                              K - (cc + 1) * BK);                                           \
        __syncthreads();                                                                    \
        cur ^= 1;                                                                           \
        ++cc;                                                                               \
    }
    const int ngroups = nch / PF, rem = nch % PF;
    for (int gi = 0; gi < ngroups; ++gi) {
        BM_STEP(r0, r1)
        BM_STEP(r1, r2)
        BM_STEP(r2, r0)
    }
    if (rem >= 1) BM_STEP(r0, r1)
    if (rem >= 2) BM_STEP(r1, r2)
#undef BM_STEP
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
__device__ __forceinline__ void block_to_tile(int tiles_j, int &ti, int &tj) {
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb / 8, r = nb % 8, xcd = b % 8;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    ti = t / tiles_j;
    tj = t % tiles_j;
}

}  // namespace bm
