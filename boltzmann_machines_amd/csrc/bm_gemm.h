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
// Geometry (wave64, gfx950), a compile-time parameter pack Geo<WI, WJ, MI, NJ, BK>:
//   workgroup = WI x WJ waves; a wave owns MI x NJ MFMA tiles of 16 x 16 outputs, i.e.
//   16*MI outputs along i (the contiguous output dim) x 16*NJ along j.
//   MFMA roles: A-operand <- P[i][k], B-operand <- Q[j][k].  With MI = 2 the two MFMA
//   tiles of a wave interleave along i (tile t holds i = base + 2m + t, m = MFMA row), so
//     * ONE ds_read_b64 feeds the A operand of both MFMAs of a k-step, and
//     * accumulator lane (l&15)=j, group g=l>>4 holds the 8 CONSECUTIVE outputs
//       i = base + 8g + 2r + t  (r = register, t = tile) of output row j:
//       two 16-byte stores and exactly two Philox blocks per lane.
//   With MI = 1 a lane holds the 4 consecutive outputs i = base + 4g + r (one Philox block).
//   K is streamed in BK chunks through a 3-slot LDS ring; global->VGPR loads run four
//   chunks ahead (the whole problem is L2/MALL resident, the loop is latency- not
//   bandwidth-limited).
//
// Operand storage:
//   P is always k-major  [k][i] (i contiguous): W for prop-up, the maintained
//     transpose Wt for prop-down, the hidden means for the outer products.
//   Q is k-major [k][j] (outer products: X, v) or x-major [j][k] (propagations:
//     rows of X / h, k contiguous).
// LDS strides make every fragment read conflict free:
//   P  MI=2 (ds_read_b64, 64 banks): stride == 32 (mod 64): lanes 0-15 cover 32
//      consecutive dwords of row k, lanes 16-31 the other 32 banks with row k+1.
//   P  MI=1 / Q KM (ds_read_b32, 32 banks): stride == 16 (mod 32).
//   Q XM (ds_read_b32, half-wave passes over 32 banks): stride BK+2 == 2 (mod 4): bank =
//      (2j + k) mod 32 is a bijection on 16 j x 2 k.  The b64 stores of an x-major tile go out
//      in passes of 16 lanes; xm_slot() gives 16 consecutive lanes the first (or second) 8
//      float4 of TWO adjacent rows, whose dwords 4c+{0,1} (+2 for the odd row) cover all 32
//      banks (16 lanes of ONE row collide 2-way: rocprofv3 SQ_LDS_BANK_CONFLICT).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bm {

// Side work hooks of the main loop: fill() runs in the shadow of the pipeline fill (the first
// global round trip), drain() is issued before the MFMAs of the last two chunks (loads the
// epilogue needs: their latency hides under ~2k cycles of matrix work).  Default: none.
struct NoSide {
    static constexpr bool kFinalSync = true;     // a following pipeline may refill the LDS ring
    __device__ __forceinline__ void fill() {}
    __device__ __forceinline__ void drain() {}
};

// compile-time ablation mask (template parameter ABL, 0 in the product; tools/probe_act.hip
// instantiates other values to price each pipeline stage)
#define BM_ABL(bit) ((ABL >> (bit)) & 1)

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 256;   // threads per workgroup of the non-tile kernels (column sums, max-norm)
constexpr int NBUF = 3;   // LDS ring depth

enum : int { KM = 0, XM = 1 };

// Tile geometry (see the header comment).  GeoAct: propagations at the north-star shape
// (batch 512 x 1024 hidden -> 16 x 16 = 256 tiles, one per CU); GeoGrad: 64 x 64 outer-product
// tiles (4 MFMAs per fragment pair, 208 tiles at 784 x 1024).
template <int WI_, int WJ_, int MI_, int NJ_, int BK_>
struct Geo {
    static constexpr int WI = WI_, WJ = WJ_, MI = MI_, NJ = NJ_, BK = BK_;
    static constexpr int NT = 64 * WI * WJ;              // threads per workgroup
    static constexpr int TI = 16 * MI * WI;              // tile extent along i
    static constexpr int TJ = 16 * NJ * WJ;              // tile extent along j
    static constexpr int E = 4 * MI;                     // consecutive outputs per lane and j sub-tile
    static constexpr int P_STRIDE = (MI == 2) ? ((TI % 64 == 0) ? TI + 32 : TI) : ((TI % 32 == 0) ? TI + 16 : TI);
    static constexpr int Q_STRIDE_XM = BK + 2;
    // k-major Q: NJ == 2 interleaves the two j sub-tiles like the i side (sub-tile n holds
    // j = base + 2*l15 + n: ONE ds_read_b64 per k-step, stride == 32 mod 64); NJ == 1 reads b32
    static constexpr int Q_STRIDE_KM = (NJ == 2) ? ((TJ % 64 == 0) ? TJ + 32 : TJ) : ((TJ % 32 == 0) ? TJ + 16 : TJ);
    // x-major P (MI == 1 only: W itself as the prop-down operand, no maintained transpose): same
    // row stride and conflict rules as the x-major Q
    static constexpr int P_STRIDE_XM = BK + 2;
    static constexpr int P_BUF = (MI == 1 && TI * P_STRIDE_XM > BK * P_STRIDE) ? TI * P_STRIDE_XM : BK * P_STRIDE;
    static constexpr int Q_BUF = (BK * Q_STRIDE_KM > TJ * Q_STRIDE_XM) ? BK * Q_STRIDE_KM : TJ * Q_STRIDE_XM;
    static constexpr int SMEM_FLOATS = NBUF * (P_BUF + Q_BUF);
    static constexpr int NVP = TI * BK / (4 * NT);       // float4 per thread per chunk, P tile
    static constexpr int NVQ = TJ * BK / (4 * NT);       // ... Q tile
    static_assert(MI == 1 || MI == 2, "MI");
    static_assert(NJ == 1 || NJ == 2, "NJ");
    static_assert(NVP >= 1 && NVP * 4 * NT == TI * BK, "P chunk must split evenly over the threads");
    static_assert(NVQ >= 1 && NVQ * 4 * NT == TJ * BK, "Q chunk must split evenly over the threads");
    static_assert(BK % 8 == 0, "BK");
};
using GeoAct = Geo<2, 2, 2, 1, 64>;      // 64 x 32 tile, 4 waves of 32 x 16, 108 KiB LDS
using GeoAct8 = Geo<2, 4, 1, 1, 64>;     // 32 x 64 tile, 8 waves of 16 x 16 (two per SIMD), 96 KiB LDS:
                                         // more LDS traffic per MFMA but half the per-wave fill and
                                         // epilogue; wins while the launch is about one tile per CU
using GeoActS = Geo<2, 2, 1, 1, 64>;     // 32 x 32 tile, 4 waves of 16 x 16, 72 KiB LDS (two per CU): for
                                         // outputs too small to give every CU a larger tile
using GeoActS32 = Geo<2, 2, 1, 1, 32>;   // the same with BK = 32: 36 KiB LDS, up to four workgroups per CU
using GeoGrad = Geo<2, 2, 2, 2, 64>;     // 64 x 64 tile, 4 waves of 32 x 32, 144 KiB LDS
using GeoGrad8 = Geo<2, 4, 2, 1, 64>;    // 64 x 64 tile, 8 waves of 32 x 16 (two per SIMD: one wave's LDS / global issue
                                         // overlaps its partner's MFMAs), 132 KiB LDS

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

// float4 slot f of an x-major tile chunk -> (row, c4); see the bank note in the header
template <int BK>
__device__ __forceinline__ void xm_slot(int f, int &row, int &c4) {
    constexpr int C = BK / 4;                  // float4 per row
    static_assert(C % 8 == 0, "xm_slot: rows of at least 8 float4");
    const int blk = f / (2 * C), r = f % (2 * C);          // two rows per block of 2*C slots
    const int q = r / 8;                                   // run of 8 lanes
    row = 2 * blk + (q & 1);
    c4 = (q >> 1) * 8 + (r & 7);
}

// global -> registers for one BK chunk of one operand tile (TX = tile extent along x,
// NTH = threads of the workgroup).
// FAST: every float4 is either fully inside or fully outside the operand (host
// guarantees 16B alignment, ld % 4 == 0 and a contiguous extent % 4 == 0), so the
// load is unconditional and branch free: indices are clamped into range; the K tail is
// zeroed when the set is stored to LDS, and x-tail garbage only reaches outputs
// i >= I / j >= J, which are never stored.
template <int L, int TX, int BK, int NTH, bool FAST>
__device__ __forceinline__ void g2r(float4 (&reg)[TX * BK / (4 * NTH)], const float *ptr, int ld, int nx, int vec,
                                    int x0, int k0, int K, int tid) {
    constexpr int NV = TX * BK / (4 * NTH);   // float4 per thread
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int f = tid + n * NTH;
        int k, x;
        if (L == KM) {
            const int row = f / (TX / 4), c4 = f % (TX / 4);
            k = k0 + row; x = x0 + c4 * 4;
        } else {
            int row, c4;
            xm_slot<BK>(f, row, c4);
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
// FAST path needs it, the guarded loads already returned zeros).  STRIDE = LDS row stride
// of this layout (k rows for KM, x rows for XM).
template <int L, int TX, int BK, int NTH, int STRIDE, bool FAST>
__device__ __forceinline__ void r2s(const float4 (&reg)[TX * BK / (4 * NTH)], float *s, int tid, int kz) {
    constexpr int NV = TX * BK / (4 * NTH);
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int f = tid + n * NTH;
        float4 v = reg[n];
        if (L == KM) {
            const int row = f / (TX / 4), c4 = f % (TX / 4);
            if (FAST && row >= kz) v = make_float4(0.f, 0.f, 0.f, 0.f);
            float2 *d = reinterpret_cast<float2 *>(s + row * STRIDE + c4 * 4);
            d[0] = make_float2(v.x, v.y);
            d[1] = make_float2(v.z, v.w);
        } else {
            int row, c4;
            xm_slot<BK>(f, row, c4);
            if (FAST && c4 * 4 >= kz) v = make_float4(0.f, 0.f, 0.f, 0.f);
            float2 *d = reinterpret_cast<float2 *>(s + row * STRIDE + c4 * 4);
            d[0] = make_float2(v.x, v.y);
            d[1] = make_float2(v.z, v.w);
        }
    }
}

// MFMA operand fragments of one BK chunk for this wave (48 VGPRs for GeoAct)
template <class G> struct Frags {
    float p[G::BK / 4][G::MI];     // p[kk][t] = P[k = 4kk+g][i = base + MI*l15 + t]
    float q[G::BK / 4][G::NJ];     // q[kk][n] = Q[j = 16n + l15][k = 4kk+g]
};

template <int QL, class G, int ABL = 0, int PL = KM>
__device__ __forceinline__ void read_frags(Frags<G> &f, const float *sP, const float *sQ, int wi, int wj, int lane) {
    static_assert(PL == KM || G::MI == 1, "x-major P needs MI == 1 (no interleaved sub-tiles)");
    const int g = lane >> 4, l15 = lane & 15;
    const float *pP = (PL == KM) ? sP + g * G::P_STRIDE + wi * (16 * G::MI) + G::MI * l15
                                 : sP + (wi * 16 + l15) * G::P_STRIDE_XM + g;
    const float *pQ = (QL == KM) ? sQ + g * G::Q_STRIDE_KM + wj * 16 * G::NJ + ((G::NJ == 2) ? 2 * l15 : l15)
                                 : sQ + (wj * 16 * G::NJ + l15) * G::Q_STRIDE_XM + g;
#pragma unroll
    for (int kk = 0; kk < G::BK / 4; ++kk) {
        if (BM_ABL(3)) { f.p[kk][0] = 1.f; if (G::MI == 2) f.p[kk][G::MI - 1] = 2.f; f.q[kk][0] = 1.f; continue; }
        // (hipcc fuses pairs of these into ds_read2st64_b64; keeping them as separate ds_read_b64
        // was measured 34 % SLOWER in the loop, so the fused form stays)
        if (G::MI == 2) {
            const float2 t = *reinterpret_cast<const float2 *>(pP + kk * 4 * G::P_STRIDE);
            f.p[kk][0] = t.x; f.p[kk][G::MI - 1] = t.y;
        } else {
            f.p[kk][0] = (PL == KM) ? pP[kk * 4 * G::P_STRIDE] : pP[kk * 4];
        }
        if (QL == KM && G::NJ == 2) {
            const float2 t = *reinterpret_cast<const float2 *>(pQ + kk * 4 * G::Q_STRIDE_KM);
            f.q[kk][0] = t.x; f.q[kk][G::NJ - 1] = t.y;
        } else {
#pragma unroll
            for (int n = 0; n < G::NJ; ++n)
                f.q[kk][n] = (QL == KM) ? pQ[kk * 4 * G::Q_STRIDE_KM + 16 * n] : pQ[16 * n * G::Q_STRIDE_XM + kk * 4];
        }
    }
}

// acc[t][n] += P-frag(t) x Q-frag(n)
template <class G, int ABL = 0>
__device__ __forceinline__ void mfma_frags(f32x4 (&acc)[G::MI][G::NJ], const Frags<G> &f) {
#pragma unroll
    for (int kk = 0; kk < G::BK / 4; ++kk) {
#pragma unroll
        for (int n = 0; n < G::NJ; ++n) {
            const float q = f.q[kk][n];
#pragma unroll
            for (int t = 0; t < G::MI; ++t) {
                if (BM_ABL(1)) { acc[t][n][0] += f.p[kk][t] * q; continue; }
                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.p[kk][t], q, acc[t][n], 0, 0, 0);
            }
        }
    }
}

// the same for the first `nq` groups of 4 k-steps only (last chunk of a contraction whose
// K tail is shorter than BK: the zero-filled remainder would only add fma(0, 0, acc))
template <class G, int ABL = 0>
__device__ __forceinline__ void mfma_frags_head(f32x4 (&acc)[G::MI][G::NJ], const Frags<G> &f, int nq) {
#pragma unroll
    for (int q = 0; q < G::BK / 16; ++q) {
        if (q < nq) {                    // wave-uniform
#pragma unroll
            for (int kk = 4 * q; kk < 4 * q + 4; ++kk) {
#pragma unroll
                for (int n = 0; n < G::NJ; ++n) {
                    const float qv = f.q[kk][n];
#pragma unroll
                    for (int t = 0; t < G::MI; ++t) {
                        if (BM_ABL(1)) { acc[t][n][0] += f.p[kk][t] * qv; continue; }
                        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.p[kk][t], qv, acc[t][n], 0, 0, 0);
                    }
                }
            }
        }
    }
}

// one register set = one BK chunk of both operand tiles, global -> VGPR staging
template <class G> struct ChunkRegs {
    float4 p[G::NVP];
    float4 q[G::NVQ];
};

// The K range of a contraction: segment 1 followed by an optional segment 2 with its own
// operands (DBM two-sided layer input; positive then negative phase of the outer
// products, the latter with a NEGATED P operand: fma(-p, q, acc) == acc - p*q exactly),
// streamed as ONE continuous pipeline (one fill, one drain).
struct KRange {
    Operand P1, Q1; int K1;
    Operand P2, Q2; int K2;     // K2 == 0: absent
};

// branch-free wave-uniform selects (a branch inside a step would split its scheduling region)
__device__ __forceinline__ int sel_i(int a, int b, int m) { return a ^ ((a ^ b) & m); }
__device__ __forceinline__ const float *sel_p(const float *a, const float *b, int m) {
    const uintptr_t ua = (uintptr_t)a, ub = (uintptr_t)b;
    return (const float *)(ua ^ ((ua ^ ub) & (uintptr_t)(intptr_t)m));
}

template <int QL, class G, bool FAST, bool SEG2, int PL = KM>
__device__ __forceinline__ void load_chunk(ChunkRegs<G> &r, const KRange &kr, int nch1, int i0, int j0, int c, int tid) {
    constexpr int BK = G::BK;
    if (!SEG2) {
        g2r<PL, G::TI, BK, G::NT, FAST>(r.p, kr.P1.ptr, kr.P1.ld, kr.P1.nx, kr.P1.vec, i0, c * BK, kr.K1, tid);
        g2r<QL, G::TJ, BK, G::NT, FAST>(r.q, kr.Q1.ptr, kr.Q1.ld, kr.Q1.nx, kr.Q1.vec, j0, c * BK, kr.K1, tid);
    } else {
        const int m = -(int)((c >= nch1) & (kr.K2 > 0));   // all-ones in segment 2 (wave-uniform)
        const int kc = c - (nch1 & m);
        const int K = sel_i(kr.K1, kr.K2, m);
        g2r<PL, G::TI, BK, G::NT, FAST>(r.p, sel_p(kr.P1.ptr, kr.P2.ptr, m), sel_i(kr.P1.ld, kr.P2.ld, m),
                                        sel_i(kr.P1.nx, kr.P2.nx, m), sel_i(kr.P1.vec, kr.P2.vec, m), i0, kc * BK, K, tid);
        g2r<QL, G::TJ, BK, G::NT, FAST>(r.q, sel_p(kr.Q1.ptr, kr.Q2.ptr, m), sel_i(kr.Q1.ld, kr.Q2.ld, m),
                                        sel_i(kr.Q1.nx, kr.Q2.nx, m), sel_i(kr.Q1.vec, kr.Q2.vec, m), j0, kc * BK, K, tid);
    }
}

template <int QL, class G, bool FAST, bool SEG2, int PL = KM>
__device__ __forceinline__ void store_chunk(const ChunkRegs<G> &r, const KRange &kr, int nch1, int c,
                                            float *sP, float *sQ, int tid) {
    constexpr int BK = G::BK;
    const int m = SEG2 ? -(int)((c >= nch1) & (kr.K2 > 0)) : 0;
    const int kz = sel_i(kr.K1, kr.K2, m) - (c - (nch1 & m)) * BK;
    r2s<PL, G::TI, BK, G::NT, (PL == KM) ? G::P_STRIDE : G::P_STRIDE_XM, FAST>(r.p, sP, tid, kz);
    r2s<QL, G::TJ, BK, G::NT, (QL == KM) ? G::Q_STRIDE_KM : G::Q_STRIDE_XM, FAST>(r.q, sQ, tid, kz);
}

// ---- steady-state ("slim") chunk path -------------------------------------------------------
// One wave issues its instructions strictly in order and an MFMA holds the wave for its 32
// cycles (tools/ubench.hip: MFMA + n VALU = 41 + 4.4 n cycles in one wave), so every VALU
// instruction inside the K loop is paid in full.  For chunks that lie completely inside K the
// loads therefore use per-thread byte offsets computed ONCE per kernel (x clamps folded in)
// on top of a wave-uniform chunk base (SGPR arithmetic), and the LDS stores skip the K-tail
// zero fill.  Chunks that touch the end of a segment go through load_chunk/store_chunk above.
template <class G> struct LoadPlan {
    uint32_t p[G::NVP], q[G::NVQ];     // byte offsets relative to the chunk base of the operand
};

template <int L, int TX, int BK, int NTH>
__device__ __forceinline__ void plan_offsets(uint32_t (&off)[TX * BK / (4 * NTH)], int ld, int nx, int x0, int tid) {
    constexpr int NV = TX * BK / (4 * NTH);
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int f = tid + n * NTH;
        if (L == KM) {
            const int row = f / (TX / 4), c4 = f % (TX / 4);
            off[n] = (uint32_t)(row * ld + min(x0 + c4 * 4, nx - 4)) * 4u;
        } else {
            int row, c4;
            xm_slot<BK>(f, row, c4);
            off[n] = (uint32_t)(min(x0 + row, nx - 1) * ld + c4 * 4) * 4u;
        }
    }
}

template <int QL, class G, int PL = KM>
__device__ __forceinline__ void make_plan(LoadPlan<G> &pl, const Operand &P, const Operand &Q, int i0, int j0, int tid) {
    plan_offsets<PL, G::TI, G::BK, G::NT>(pl.p, P.ld, P.nx, i0, tid);
    plan_offsets<QL, G::TJ, G::BK, G::NT>(pl.q, Q.ld, Q.nx, j0, tid);
}

// chunk c (must be a FULL chunk of its segment; c >= nch: prefetch overrun, any full chunk does)
template <int QL, class G, bool SEG2, int PL = KM>
__device__ __forceinline__ void load_chunk_slim(ChunkRegs<G> &r, const KRange &kr, const LoadPlan<G> &pl1,
                                                const LoadPlan<G> &pl2, int nch1, int nch, int c) {
    constexpr int BK = G::BK;
    const int cl = (c < nch) ? c : 0;
    const int m = SEG2 ? -(int)(cl >= nch1) : 0;          // all-ones in segment 2 (wave-uniform)
    const int kc = cl - (nch1 & m);
    const int ldp = SEG2 ? sel_i(kr.P1.ld, kr.P2.ld, m) : kr.P1.ld;
    const int ldq = SEG2 ? sel_i(kr.Q1.ld, kr.Q2.ld, m) : kr.Q1.ld;
    const char *pb = (const char *)((SEG2 ? sel_p(kr.P1.ptr, kr.P2.ptr, m) : kr.P1.ptr) +
                                    ((PL == KM) ? (size_t)kc * BK * ldp : (size_t)kc * BK));
    const char *qb = (const char *)((SEG2 ? sel_p(kr.Q1.ptr, kr.Q2.ptr, m) : kr.Q1.ptr) +
                                    ((QL == KM) ? (size_t)kc * BK * ldq : (size_t)kc * BK));
#pragma unroll
    for (int n = 0; n < G::NVP; ++n) {
        const uint32_t o1 = pl1.p[n], o2 = SEG2 ? pl2.p[n] : 0u;
        const uint32_t o = SEG2 ? (o1 ^ ((o1 ^ o2) & (uint32_t)m)) : o1;
        r.p[n] = *reinterpret_cast<const float4 *>(pb + o);
    }
#pragma unroll
    for (int n = 0; n < G::NVQ; ++n) {
        const uint32_t o1 = pl1.q[n], o2 = SEG2 ? pl2.q[n] : 0u;
        const uint32_t o = SEG2 ? (o1 ^ ((o1 ^ o2) & (uint32_t)m)) : o1;
        r.q[n] = *reinterpret_cast<const float4 *>(qb + o);
    }
}

template <int QL, class G, int PL = KM>
__device__ __forceinline__ void store_chunk_slim(const ChunkRegs<G> &r, float *sP, float *sQ, int tid) {
    r2s<PL, G::TI, G::BK, G::NT, (PL == KM) ? G::P_STRIDE : G::P_STRIDE_XM, false>(r.p, sP, tid, 0);
    r2s<QL, G::TJ, G::BK, G::NT, (QL == KM) ? G::Q_STRIDE_KM : G::Q_STRIDE_XM, false>(r.q, sQ, tid, 0);
}

// acc += sum_k P[k][i] * Q[j][k] over the K range, k ascending (canonical order).
//
// The loop is software-pipelined by hand over a 3-deep LDS ring.  In step c (between
// barriers B(c-1) and B(c)) a wave
//   * runs the MFMAs of chunk c on fragments F(c) that are ALREADY in registers,
//   * reads the fragments F(c+1) from LDS slot (c+1)%3 (published by B(c-1)),
//   * stores chunk c+2 (global data that arrived in registers) to LDS slot (c+2)%3
//     (last read for F(c-1), complete before B(c-2)),
//   * re-issues the global loads of chunk c+4 into the register set just stored.
// sched_group_barrier pins that issue order: hipcc otherwise issues the LDS traffic
// AFTER the MFMAs and the two phases run back to back.  Steps come in two flavours, chosen
// per PAIR of steps by a wave-uniform branch around the whole pair (so each flavour is one
// scheduling region): the slim one for full chunks (see LoadPlan) and the careful one
// (clamped loads, K-tail zero fill) for chunks that touch the end of a segment.
// `side.fill()` (the lane's Philox blocks in act_kernel) runs while the first loads are in
// flight, when the wave would otherwise idle for one memory round trip.
template <int QL, class G, bool FAST, bool SEG2, int ABL = 0, int PL = KM, class Side = NoSide>
__device__ __forceinline__ void mainloop(f32x4 (&acc)[G::MI][G::NJ], const KRange &kr, int i0, int j0, float *smem,
                                         Side &side, long long *stamps = nullptr) {
#ifdef BM_PROBE
#define BM_MSTAMP(n) do { if (stamps && threadIdx.x == 0) stamps[n] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define BM_MSTAMP(n) do {} while (0)
#endif
    constexpr int BK = G::BK, P_BUF = G::P_BUF, Q_BUF = G::Q_BUF;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w % G::WI, wj = w / G::WI;
    float *sP = smem, *sQ = smem + NBUF * P_BUF;
    const int nch1 = (kr.K1 + BK - 1) / BK;
    const int nch = nch1 + (SEG2 ? (kr.K2 + BK - 1) / BK : 0);
    const int nfull1 = kr.K1 / BK, nfull2 = SEG2 ? kr.K2 / BK : 0;
    ChunkRegs<G> g0, g1;      // named sets (never arrays: must stay in VGPRs)
    Frags<G> fa, fb;
    LoadPlan<G> pl1, pl2;
    if (FAST) {
        make_plan<QL, G, PL>(pl1, kr.P1, kr.Q1, i0, j0, tid);
        if (SEG2) make_plan<QL, G, PL>(pl2, kr.P2, kr.Q2, i0, j0, tid);
    }
    BM_MSTAMP(0);
    {   // pipeline fill: the four chunk loads go out back to back (ONE memory round trip);
        // chunks 0/1 pass through two prologue-only sets, chunks 2/3 land in the loop's sets
        ChunkRegs<G> ga, gb;
        load_chunk<QL, G, FAST, SEG2, PL>(ga, kr, nch1, i0, j0, 0, tid);
        load_chunk<QL, G, FAST, SEG2, PL>(gb, kr, nch1, i0, j0, 1, tid);
        load_chunk<QL, G, FAST, SEG2, PL>(g0, kr, nch1, i0, j0, 2, tid);
        load_chunk<QL, G, FAST, SEG2, PL>(g1, kr, nch1, i0, j0, 3, tid);
        __builtin_amdgcn_sched_barrier(0);
        side.fill();
        __builtin_amdgcn_sched_barrier(0);
        store_chunk<QL, G, FAST, SEG2, PL>(ga, kr, nch1, 0, sP, sQ, tid);
        store_chunk<QL, G, FAST, SEG2, PL>(gb, kr, nch1, 1, sP + P_BUF, sQ + Q_BUF, tid);
    }
    BM_MSTAMP(1);
    __syncthreads();
    read_frags<QL, G, ABL, PL>(fa, sP, sQ, wi, wj, lane);
    BM_MSTAMP(2);
    int cc = 0, b1 = 1, b2 = 2;   // LDS slots of chunk cc+1 / cc+2
    // Issue order of one step.  Instruction counts per wave and step:
    //   NM MFMAs; NRG fused fragment-read groups of (1 + NJ) DS reads (hipcc pairs the reads of
    //   two k-steps into ds_read2st64); NW DS writes; NL global loads.
    // masks: 0x008 MFMA, 0x100 DS read, 0x200 DS write, 0x020 VMEM read
    constexpr int NM = (BK / 4) * G::MI * G::NJ;
    constexpr int NRG = BK / 8, RPG = (QL == KM && G::NJ == 2) ? 2 : 1 + G::NJ;
    constexpr int NW = ((PL == XM) ? 2 * G::NVP : G::NVP) + ((QL == XM) ? 2 * G::NVQ : G::NVQ);
    constexpr int NL = G::NVP + G::NVQ;
    constexpr int MR = (NM >= 64) ? 2 : 1;            // MFMAs per read group
    constexpr int MW = (NM >= 32) ? 2 : 1;            // MFMAs per DS write
    constexpr int REST = NM - (NRG * MR + NW * MW + NL);
#define BM_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0);
#ifndef BM_SCHED_VARIANT
#define BM_SCHED_VARIANT 0
#endif
#if BM_SCHED_VARIANT == 0
#define BM_SCHED_STEP                                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < NRG; ++s_) { BM_SG(0x008, MR) BM_SG(0x100, RPG) }    \
    _Pragma("unroll") for (int s_ = 0; s_ < NW; ++s_) { BM_SG(0x008, MW) BM_SG(0x200, 1) }        \
    _Pragma("unroll") for (int s_ = 0; s_ < NL; ++s_) { BM_SG(0x008, 1) BM_SG(0x020, 1) }         \
    if (REST > 0) { BM_SG(0x008, (REST > 0 ? REST : 1)) }
#elif BM_SCHED_VARIANT == 1    /* every memory op behind its own MFMA: reads, writes, loads */
#define BM_SCHED_STEP                                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < NRG * RPG; ++s_) { BM_SG(0x008, 1) BM_SG(0x100, 1) } \
    _Pragma("unroll") for (int s_ = 0; s_ < NW; ++s_) { BM_SG(0x008, 1) BM_SG(0x200, 1) }         \
    _Pragma("unroll") for (int s_ = 0; s_ < NL; ++s_) { BM_SG(0x008, 1) BM_SG(0x020, 1) }         \
    BM_SG(0x008, NM)
#elif BM_SCHED_VARIANT == 2    /* loads, writes, then reads, 1:1 */
#define BM_SCHED_STEP                                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < NL; ++s_) { BM_SG(0x008, 1) BM_SG(0x020, 1) }         \
    _Pragma("unroll") for (int s_ = 0; s_ < NW; ++s_) { BM_SG(0x008, 1) BM_SG(0x200, 1) }         \
    _Pragma("unroll") for (int s_ = 0; s_ < NRG * RPG; ++s_) { BM_SG(0x008, 1) BM_SG(0x100, 1) } \
    BM_SG(0x008, NM)
#elif BM_SCHED_VARIANT == 3    /* writes and loads paired 1:1 first, then reads 1:1 */
#define BM_SCHED_STEP                                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < NW; ++s_) { BM_SG(0x008, 1) BM_SG(0x200, 1) BM_SG(0x020, 1) } \
    _Pragma("unroll") for (int s_ = 0; s_ < NRG * RPG; ++s_) { BM_SG(0x008, 1) BM_SG(0x100, 1) } \
    BM_SG(0x008, NM)
#elif BM_SCHED_VARIANT == 4    /* reads 1 per 2 MFMAs with a write or load in the other slot */
#define BM_SCHED_STEP                                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < NW; ++s_) { BM_SG(0x008, 1) BM_SG(0x100, 1) BM_SG(0x008, 1) BM_SG(0x200, 1) } \
    _Pragma("unroll") for (int s_ = 0; s_ < NL; ++s_) { BM_SG(0x008, 1) BM_SG(0x100, 1) BM_SG(0x008, 1) BM_SG(0x020, 1) } \
    _Pragma("unroll") for (int s_ = 0; s_ < NRG * RPG; ++s_) { BM_SG(0x008, 1) BM_SG(0x100, 1) } \
    BM_SG(0x008, NM)
#elif BM_SCHED_VARIANT == 5    /* no pinning: leave the order to hipcc */
#define BM_SCHED_STEP
#endif
#define BM_STEP_TAIL                                                                              \
        BM_SCHED_STEP                                                                             \
        if (!BM_ABL(5)) __syncthreads();                                                          \
        b1 = b2;                                                                                  \
        b2 = (b2 == NBUF - 1) ? 0 : b2 + 1;                                                       \
        ++cc;
    // careful step: clamped loads, K-tail zero fill
#define BM_STEP(FC, FN, G_)                                                                       \
    {                                                                                             \
        if (!BM_ABL(3)) read_frags<QL, G, ABL, PL>(FN, sP + b1 * P_BUF, sQ + b1 * Q_BUF, wi, wj, lane); \
        if (!BM_ABL(2)) store_chunk<QL, G, FAST, SEG2, PL>(G_, kr, nch1, cc + 2, sP + b2 * P_BUF, sQ + b2 * Q_BUF, tid); \
        if (!BM_ABL(0)) load_chunk<QL, G, FAST, SEG2, PL>(G_, kr, nch1, i0, j0, cc + 4, tid);         \
        mfma_frags<G, ABL>(acc, FC);                                                              \
        BM_STEP_TAIL                                                                              \
    }
    // slim step: chunks cc+2 (stored) and cc+4 (loaded) are full chunks
#ifdef BM_PROBE
    // ABL bit 7: phase-ordered step (reads | stores | loads | MFMAs | barrier) with a cycle stamp
    // between the phases of step 4 -> where does one wave's issue time go?
    long long pst[6] = {0, 0, 0, 0, 0, 0};
#define BM_PSTAMP(k) if (BM_ABL(7)) { __builtin_amdgcn_sched_barrier(0); if (cc == 4) pst[k] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#else
#define BM_PSTAMP(k)
#endif
#define BM_STEP_SLIM(FC, FN, G_)                                                                  \
    {                                                                                             \
        BM_PSTAMP(0)                                                                              \
        if (!BM_ABL(3)) read_frags<QL, G, ABL, PL>(FN, sP + b1 * P_BUF, sQ + b1 * Q_BUF, wi, wj, lane); \
        BM_PSTAMP(1)                                                                              \
        if (!BM_ABL(2)) store_chunk_slim<QL, G, PL>(G_, sP + b2 * P_BUF, sQ + b2 * Q_BUF, tid);       \
        BM_PSTAMP(2)                                                                              \
        if (!BM_ABL(0)) load_chunk_slim<QL, G, SEG2, PL>(G_, kr, pl1, pl2, nch1, nch, cc + 4);        \
        BM_PSTAMP(3)                                                                              \
        mfma_frags<G, ABL>(acc, FC);                                                              \
        BM_PSTAMP(4)                                                                              \
        if (!BM_ABL(7)) { BM_SCHED_STEP }                                                         \
        if (!BM_ABL(5)) __syncthreads();                                                          \
        BM_PSTAMP(5)                                                                              \
        b1 = b2;                                                                                  \
        b2 = (b2 == NBUF - 1) ? 0 : b2 + 1;                                                       \
        ++cc;                                                                                     \
    }
    // steady state: steps 0 .. nch-3 carry the full LDS / global traffic; the last two steps
    // (pipeline drain) only have fragments to read and MFMAs to issue
    const int full = (nch > 2) ? nch - 2 : 0;
    const int npairs = full / 2;
    // can the pair of steps starting at step c0 take the slim path?  chunks c0+2, c0+3 (stored)
    // and c0+4, c0+5 (loaded; past nch: prefetch overrun) must all be full chunks
    auto pair_is_slim = [&](int c0) -> bool {
        if (!FAST || BM_ABL(6) || nfull1 == 0) return false;      // (overrun loads fall back on chunk 0)
        bool ok = true;
#pragma unroll
        for (int d = 2; d < 6; ++d) {
            const int c = c0 + d;
            ok = ok && ((c < nfull1) | (SEG2 && c >= nch1 && c - nch1 < nfull2) | (d >= 4 && c >= nch));
        }
        return ok;
    };
    // Runs of slim / careful pairs as separate COUNTED loops (an if/else inside one loop, or a
    // loop whose exit test depends on the chunk state, makes hipcc copy ~100 loop-carried
    // registers - and wait for the loads in flight - on every back edge).
    auto run_length = [&](int c0, int remaining, bool want_slim) -> int {
        int n = 0;
        while (n < remaining && pair_is_slim(c0 + 2 * n) == want_slim) ++n;
        return n;
    };
    int left = npairs;
#pragma unroll 1
    for (int round = 0; round < 2 && left > 0; ++round) {       // segment 1, then segment 2
        const int ns = run_length(cc, left, true);
#pragma unroll 1
        for (int q = 0; q < ns; ++q) {
            BM_STEP_SLIM(fa, fb, g0)
            BM_STEP_SLIM(fb, fa, g1)
        }
        left -= ns;
        const int nc = (round == 1) ? left : run_length(cc, left, false);
#pragma unroll 1
        for (int q = 0; q < nc; ++q) {
            BM_STEP(fa, fb, g0)
            BM_STEP(fb, fa, g1)
        }
        left -= nc;
    }
    if (full & 1) {
        BM_STEP(fa, fb, g0)
        fa = fb;               // keep the current fragments in `fa` for the drain (once per kernel)
    }
    BM_MSTAMP(3);
    // groups of 16 k that the last chunk really holds
    const int klast = (SEG2 && kr.K2 > 0) ? kr.K2 - (nch - nch1 - 1) * BK : kr.K1 - (nch1 - 1) * BK;
    const int nq_last = (klast + 15) / 16;
    side.drain();
    if (nch >= 2) {
        if (!BM_ABL(3)) read_frags<QL, G, ABL, PL>(fb, sP + b1 * P_BUF, sQ + b1 * Q_BUF, wi, wj, lane);
        mfma_frags<G, ABL>(acc, fa);
        mfma_frags_head<G, ABL>(acc, fb, nq_last);
    } else {
        mfma_frags_head<G, ABL>(acc, fa, nq_last);
    }
    if (Side::kFinalSync) __syncthreads();     // the LDS ring may be refilled by a following pipeline
    BM_MSTAMP(4);
#ifdef BM_PROBE
    if (BM_ABL(7) && stamps && lane == 0 && (w == 0 || w == 3)) {
        long long *o = stamps + 2048 + (long long)blockIdx.x * 8 + (w ? 8 : 0);    // (stamps = dbg + 2048 + 8*block)
        for (int k = 0; k < 6; ++k) o[k] = pst[k];
    }
#undef BM_PSTAMP
#endif
#undef BM_STEP
#undef BM_STEP_SLIM
#undef BM_STEP_TAIL
#undef BM_SCHED_STEP
#undef BM_SG
}

// j coordinate (within the wave's 16*NJ columns) of a lane's outputs in j sub-tile n.
// Must match read_frags: k-major Q with NJ == 2 interleaves the sub-tiles.
template <int QL, class G>
__device__ __forceinline__ int lane_j(int l15, int n) {
    return (QL == KM && G::NJ == 2) ? 2 * l15 + n : 16 * n + l15;
}

// the E = 4*MI consecutive outputs of a lane for j sub-tile n: v[e], e = MI*r + t  <->  i = ib + e
template <class G>
__device__ __forceinline__ void lane_outputs(const f32x4 (&acc)[G::MI][G::NJ], int n, float (&v)[G::E]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < G::MI; ++t) v[G::MI * r + t] = acc[t][n][r];
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
