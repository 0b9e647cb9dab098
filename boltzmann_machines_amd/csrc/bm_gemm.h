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
//   4 waves = 2 (i) x 2 (j); a wave owns 32 i x 16 j = two 16x16 MFMA tiles
//   MFMA roles: A-operand <- P[i][k], B-operand <- Q[j][k]; accumulator lane
//   (l&15)=j, regs r=0..3 -> i=(l>>4)*4+r : each lane holds 4 CONSECUTIVE i of
//   one output row j = one 16-byte store and exactly one Philox block.
//   K is streamed in BK=32 chunks through double-buffered LDS with register
//   prefetch (global->VGPR for chunk c+1 is in flight while chunk c computes).
//
// Operand storage layouts (both appear in the reference because W is used as
// W, W^T and the outer products contract over the batch):
//   KM : [k][x]  x contiguous  (W for prop-up; X, H for the outer products)
//   XM : [x][k]  k contiguous  (X/H rows for propagations; W for prop-down)
// LDS strides are padded so that ds_read_b32 fragment reads are conflict free:
//   KM stride == 16 (mod 32): lanes 0-15 read 16 consecutive dwords of row k,
//                             lanes 16-31 the same columns of row k+1.
//   XM stride == 2  (mod 4) : bank = (2x + k) mod 32 is a bijection on 16x2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TI = 64;    // tile extent along i
constexpr int TJ = 32;    // tile extent along j
constexpr int BK = 32;    // K chunk
constexpr int NT = 256;   // threads per workgroup

enum : int { KM = 0, XM = 1 };

constexpr int P_STRIDE_KM = TI + 16;   // 80
constexpr int Q_STRIDE_KM = TJ + 16;   // 48
constexpr int STRIDE_XM   = BK + 2;    // 34
constexpr int P_BUF = (BK * P_STRIDE_KM > TI * STRIDE_XM) ? BK * P_STRIDE_KM : TI * STRIDE_XM;  // 2560
constexpr int Q_BUF = (BK * Q_STRIDE_KM > TJ * STRIDE_XM) ? BK * Q_STRIDE_KM : TJ * STRIDE_XM;  // 1536
constexpr int SMEM_FLOATS = 2 * (P_BUF + Q_BUF);                                                 // 32 KiB

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

// global -> registers for one BK chunk of one operand tile (TX = TI or TJ)
template <int L, int TX>
__device__ __forceinline__ void g2r(float4 (&reg)[TX / 32], const Operand &op, int x0, int k0, int K, int tid) {
    constexpr int NV = TX / 32;   // float4 per thread
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int f = tid + n * NT;
        if (L == KM) {
            const int row = f / (TX / 4), c4 = f % (TX / 4);
            const int k = k0 + row, x = x0 + c4 * 4;
            reg[n] = load4_guard(op.ptr + (size_t)k * op.ld + x, k < K, x, op.nx, op.vec);
        } else {
            const int row = f / (BK / 4), c4 = f % (BK / 4);
            const int x = x0 + row, k = k0 + c4 * 4;
            reg[n] = load4_guard(op.ptr + (size_t)x * op.ld + k, x < op.nx, k, K, op.vec);
        }
    }
}

// registers -> LDS
template <int L, int TX>
__device__ __forceinline__ void r2s(const float4 (&reg)[TX / 32], float *s, int tid) {
    constexpr int NV = TX / 32;
    constexpr int STRIDE_K = (TX == TI) ? P_STRIDE_KM : Q_STRIDE_KM;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int f = tid + n * NT;
        if (L == KM) {
            const int row = f / (TX / 4), c4 = f % (TX / 4);
            *reinterpret_cast<float4 *>(s + row * STRIDE_K + c4 * 4) = reg[n];
        } else {
            const int row = f / (BK / 4), c4 = f % (BK / 4);
            float2 *d = reinterpret_cast<float2 *>(s + row * STRIDE_XM + c4 * 4);
            d[0] = make_float2(reg[n].x, reg[n].y);
            d[1] = make_float2(reg[n].z, reg[n].w);
        }
    }
}

// one BK chunk of MFMAs for this wave: acc[t] (t = i-subtile) += P-frag x Q-frag
template <int PL, int QL>
__device__ __forceinline__ void compute_chunk(f32x4 (&acc)[2], const float *sP, const float *sQ,
                                              int wi, int wj, int lane) {
    const int g = lane >> 4, l15 = lane & 15;
    const float *pP = (PL == KM) ? sP + g * P_STRIDE_KM + wi * 32 + l15
                                 : sP + (wi * 32 + l15) * STRIDE_XM + g;
    const float *pQ = (QL == KM) ? sQ + g * Q_STRIDE_KM + wj * 16 + l15
                                 : sQ + (wj * 16 + l15) * STRIDE_XM + g;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
        const float q  = (QL == KM) ? pQ[kk * 4 * Q_STRIDE_KM] : pQ[kk * 4];
        const float p0 = (PL == KM) ? pP[kk * 4 * P_STRIDE_KM]      : pP[kk * 4];
        const float p1 = (PL == KM) ? pP[kk * 4 * P_STRIDE_KM + 16] : pP[16 * STRIDE_XM + kk * 4];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p0, q, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p1, q, acc[1], 0, 0, 0);
    }
}

// acc[t][r] += sum_k P[i][k] * Q[j][k] over k in [0, K), k ascending (canonical order)
template <int PL, int QL>
__device__ __forceinline__ void mainloop(f32x4 (&acc)[2], const Operand &P, const Operand &Q, int K,
                                         int i0, int j0, float *smem) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w & 1, wj = w >> 1;
    float *sP[2] = {smem, smem + P_BUF};
    float *sQ[2] = {smem + 2 * P_BUF, smem + 2 * P_BUF + Q_BUF};
    const int nch = (K + BK - 1) / BK;
    float4 rp[TI / 32], rq[TJ / 32];
    g2r<PL, TI>(rp, P, i0, 0, K, tid);
    g2r<QL, TJ>(rq, Q, j0, 0, K, tid);
    r2s<PL, TI>(rp, sP[0], tid);
    r2s<QL, TJ>(rq, sQ[0], tid);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const int cur = c & 1;
        if (c + 1 < nch) {
            g2r<PL, TI>(rp, P, i0, (c + 1) * BK, K, tid);
            g2r<QL, TJ>(rq, Q, j0, (c + 1) * BK, K, tid);
        }
        compute_chunk<PL, QL>(acc, sP[cur], sQ[cur], wi, wj, lane);
        if (c + 1 < nch) {
            r2s<PL, TI>(rp, sP[cur ^ 1], tid);
            r2s<QL, TJ>(rq, sQ[cur ^ 1], tid);
        }
        __syncthreads();
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
