// bm_gemm.h — the fp32-MFMA tile engine every hot kernel of the engine is built on.
//
// All contractions of the reference hot path (tf.matmul at base_rbm.py:329-337,
// :447-448; dbm.py:390-425, :553-570, :650-694) are small dense fp32 GEMMs whose
// result feeds a nonlinearity + a Bernoulli draw or a parameter update.  They are
// computed with v_mfma_f32_16x16x4_f32 (exact fp32: the instruction is the fma chain
// acc = fmaf(a[g], b[g], acc) for g = 0..3 over the four k the lanes' 16-groups hold).
//
// "Canonical order"  — the order in which the products of one dot product
// enter the chain, identical in oracle/bm_oracle.c, so that every result is BIT-IDENTICAL
// between the CPU oracle and the GPU:
//     for each aligned block of 16 k (m = 0, 1, ...):  for j = 0..3:  for g = 0..3:
//         k = 16 m + 4 g + j;   acc = fmaf(p[k], q[k], acc)            (k >= K skipped)
// i.e. MFMA step j of block m takes k = 16m + 4g + j from lane group g.  (Round 1 used plain
// ascending k, which needs lane group g to hold k = 4 step + g: four DIFFERENT 16-byte chunks
// of a k-contiguous row per lane.  With this order a lane's four steps are ONE 16-byte chunk
// (k = 16m + 4g + {0,1,2,3}): one ds_read_b128 instead of four ds_read_b32, at twice the LDS
// rate, and the chunk can be placed in LDS by the LDS-DMA engine, which moves whole 16-byte
// pieces and cannot transpose within them.)  Segment 2 of a two-segment contraction starts
// its own block structure at its k = 0.
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
//
// Data movement: K is streamed in BK chunks through a 4-slot LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4: global -> LDS without passing through VGPRs; 1 KiB per wave
// instruction, destination = wave-uniform base + 16 * lane).  The probe (tools/probe_act.hip)
// priced round 1's VGPR -> ds_write staging at 2.0 of 13.4 us per propagation kernel and 1.9 of
// 25.3 us per outer-product kernel: the LDS write port ingests ds_write data at ~70 B/clk, and
// those cycles come out of the same pipe the fragment reads use.  DMA runs three chunks ahead
// (counted s_waitcnt vmcnt + raw s_barrier: a __syncthreads() would drain the DMA queue).
//
// Issue discipline: a wave's MFMAs form one dependent chain, and a vector-ALU instruction between two of them costs
// ~13 cycles of matrix-pipe time (tools/ubench_step.hip; scalar instructions and the DMA issue are nearly free).
// Every loop is therefore unrolled by the ring depth so that the ring slot is a compile-time constant - all LDS
// addresses become loop-invariant bases + immediates - and the DMA addresses a scalar chunk base plus a fixed
// per-lane offset: the steady steps contain no VALU instruction at all.
//
// LDS images (no padding: a DMA piece is 1 KiB of consecutive LDS; bank conflicts are avoided
// by XOR-swizzling WHICH global 16-byte chunk a lane fetches — tools/bank_check.py proves every
// fragment read conflict free under the bank model of MI355X_MICROARCH.md):
//   x-major tile [x rows][BK floats]  (propagation Q operands, W itself as the prop-down P):
//       chunk c of row r sits at slot c ^ fx(r), fx(r) = r & 15 (BK = 64) | (r >> 1) & 7 (BK = 32);
//       fragment of block m: ONE ds_read_b128 of chunk 4m + g.
//   k-major tile [BK rows k][TX floats]  (W for the prop-up, every outer-product operand):
//       chunk c of row k sits at slot c ^ (S * ((k >> 2) & 1)),  S = 4 (b32 reads) | 8 (b64 reads);
//       fragment of step (m, j): row k = 16m + 4g + j, b32 (one sub-tile) or b64 (two interleaved).
// Chunks that touch the end of a K segment, and every chunk of shapes that do not allow 16-byte
// loads (template FAST = false), pass through registers (clamped / guarded loads, zero fill) and are
// written to the SAME images with ds_write_b128.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

namespace bm {

// Side work hooks of the main loop: fill() runs in the shadow of the pipeline fill (the first
// global round trip), drain() is issued at the start of the last step but one (loads the
// epilogue needs).  Default: none.
struct NoSide {
    static constexpr bool kFinalSync = true;     // a following pipeline may refill the LDS ring
    // chained launches (bm_chain.h): the Q operand was written by OTHER workgroups of the same launch.  kSplitFill: the
    // pipeline fill requests the P pieces first, calls wait_inputs() (spin on the producers' flags), then the Q pieces;
    // kCohQ: every Q load bypasses the CU's vector L1 (sc1), which may hold the previous contents of those lines.
    static constexpr bool kSplitFill = false, kCohQ = false;
    // kCanAbort (act_kernel's mean-field flavour): post_fill(), called behind the barrier that ends the pipeline fill, may set
    // `aborted` (workgroup-uniform): the main loop then waits for its own DMA pieces and returns at once
    static constexpr bool kCanAbort = false;
    __device__ __forceinline__ void wait_inputs() {}
    __device__ __forceinline__ void post_fill() {}
    __device__ __forceinline__ void fill() {}
    __device__ __forceinline__ void drain() {}
};

// compile-time ablation mask (template parameter ABL, 0 in the product; tools/probe_act.hip
// instantiates other values to price each pipeline stage)
//   bit 0: no global -> LDS traffic   bit 1: VALU instead of MFMA   bit 3: no fragment reads
//   bit 4: no epilogue (act_kernel)   bit 5: no barriers             bit 6: register path for every chunk
//   bit 8: no wait for the DMA in the steady steps (wrong results: prices the data-arrival stalls)
//   bit 9: only every other Q piece is moved (wrong results: what would half the Q bytes - a 16-bit state operand - buy?)
#define BM_ABL(bit) ((ABL >> (bit)) & 1)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

constexpr int NT = 256;        // threads per workgroup of the non-tile kernels (column sums, max-norm)

enum : int { KM = 0, XM = 1 };

// Tile geometry (see the header comment).
template <int WI_, int WJ_, int MI_, int NJ_, int BK_, int NB_ = 4>
struct Geo {
    static constexpr int WI = WI_, WJ = WJ_, MI = MI_, NJ = NJ_, BK = BK_;
    static constexpr int NBUF = NB_;                     // LDS ring depth (chunks)
    static constexpr int PF = NB_ - 1;                   // DMA prefetch distance in chunks
    static constexpr int NW = WI * WJ;                   // waves per workgroup
    static constexpr int NT = 64 * NW;                   // threads per workgroup
    static constexpr int TI = 16 * MI * WI;              // tile extent along i
    static constexpr int TJ = 16 * NJ * WJ;              // tile extent along j
    static constexpr int E = 4 * MI;                     // consecutive outputs per lane and j sub-tile
    static constexpr int P_BUF = TI * BK;                // floats of one P / Q tile chunk (no padding)
    static constexpr int Q_BUF = TJ * BK;
    static constexpr int SMEM_FLOATS = NBUF * (P_BUF + Q_BUF);
    static constexpr int NPP = P_BUF * 4 / (1024 * NW);  // 1 KiB DMA pieces per wave and chunk, P tile
    static constexpr int NPQ = Q_BUF * 4 / (1024 * NW);  // ... Q tile
    static constexpr int NVP = TI * BK / (4 * NT);       // float4 per thread per chunk (register path), P tile
    static constexpr int NVQ = TJ * BK / (4 * NT);       // ... Q tile
    static constexpr int SP = (MI == 2) ? 8 : 4;         // k-major swizzle strides (see header)
    static constexpr int SQ = (NJ == 2) ? 8 : 4;
    static_assert(MI == 1 || MI == 2, "MI");
    static_assert(NJ == 1 || NJ == 2, "NJ");
    static_assert(BK == 32 || BK == 64, "BK");
    static_assert(NB_ >= 4 && NB_ <= 8, "ring depth");
    static_assert(NPP >= 1 && NPP * 1024 * NW == P_BUF * 4, "P chunk must split into whole DMA pieces per wave");
    static_assert(NPQ >= 1 && NPQ * 1024 * NW == Q_BUF * 4, "Q chunk must split into whole DMA pieces per wave");
    static_assert(NVP >= 1 && NVP * 4 * NT == TI * BK && NVQ >= 1 && NVQ * 4 * NT == TJ * BK, "register path split");
    static_assert(SMEM_FLOATS * 4 <= 160 * 1024 - 1024, "LDS");
};
using GeoAct = Geo<2, 2, 2, 1, 64>;      // 64 x 32 tile, 4 waves of 32 x 16, 96 KiB LDS
// (Geo<2, 4, 1, 1, 64, 6>, the 8-wave tile with a 6-deep ring = DMA five chunks ahead, measured 16.2 us against 14.7
//  at 784x1024x512 and 295 against 259 us at 20000 rows: more loads in flight do not help, latency is not the bound)
using GeoAct8 = Geo<2, 4, 1, 1, 64>;     // 32 x 64 tile, 8 waves of 16 x 16 (two per SIMD), 96 KiB LDS:
                                         // more LDS traffic per MFMA but half the per-wave fill and
                                         // epilogue; wins while the launch is about one tile per CU
using GeoActS = Geo<2, 2, 1, 1, 64>;     // 32 x 32 tile, 4 waves of 16 x 16, 64 KiB LDS (two per CU): for
                                         // outputs too small to give every CU a larger tile
using GeoActS32 = Geo<2, 2, 1, 1, 32>;   // the same with BK = 32: 32 KiB LDS, up to four workgroups per CU
using GeoAct32 = Geo<2, 2, 2, 1, 32>;     // 64 x 32 tile, 4 waves of 32 x 16, BK = 32: 48 KiB LDS, three workgroups per CU;
                                         // 3/4 of the operand bytes per flop of the 32 x 32 tile (k-major P only)
using GeoGrad = Geo<2, 2, 2, 2, 64>;     // 64 x 64 tile, 4 waves of 32 x 32, 128 KiB LDS
using GeoGrad8 = Geo<2, 4, 2, 1, 64>;    // 64 x 64 tile, 8 waves of 32 x 16 (two per SIMD), 128 KiB LDS
using GeoGrad8h = Geo<2, 4, 2, 1, 32>;   // the same with BK = 32: 64 KiB LDS, two workgroups per CU (one's epilogue under the other's K loop)
using GeoBf3S = Geo<1, 4, 2, 1, 64>;     // 32 x 64 tile, 4 waves of 32 x 16: the fast-binary path's two-workgroups-per-CU tile (bm_bf3.h)

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

// host: can this operand take the 16-byte (DMA) path?
static inline bool operand_fast(const Operand &o, int layout, int K) {
    if (!o.ptr) return true;                      // absent segment
    if (!o.vec) return false;
    return layout == KM ? (o.nx % 4 == 0 && o.nx >= 4) : (K % 4 == 0);
}

// ---- LDS image addressing -------------------------------------------------------------------
template <int BK> __device__ __forceinline__ int fx(int row) { return (BK == 64) ? (row & 15) : ((row >> 1) & 7); }
template <int S> __device__ __forceinline__ int fk(int row) { return S * ((row >> 2) & 1); }

// float offset of 16-byte chunk c4 of tile row `row`; L = layout, TX = floats per k-major row, S = its swizzle
template <int L, int TX, int BK, int S>
__device__ __forceinline__ int img_off(int row, int c4) {
    return (L == XM) ? row * BK + ((c4 ^ fx<BK>(row)) << 2) : row * TX + ((c4 ^ fk<S>(row)) << 2);
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

// ---- register path (chunks that touch the end of a segment; every chunk when !FAST) ------------
// global -> registers for one BK chunk of one operand tile.  float4 slot f = tid + n*NTH covers tile row
// f / (row float4s), chunk f % (row float4s).  FAST: every float4 is either fully inside or fully outside
// the operand, so the load is unconditional with clamped indices; the K tail is zeroed at the LDS store,
// x-tail garbage only reaches outputs i >= I / j >= J, which are never stored.
// 16 bytes through the L2, past the CU's vector L1: two relaxed agent-scope loads (global_load_dwordx2 ... sc1)
__device__ __forceinline__ float4 load4_coh(const float *p) {
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)),
                       __uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32)));
}

template <int L, int TX, int BK, int NTH, bool FAST, bool COH = false>
__device__ __forceinline__ void g2r(float4 (&reg)[TX * BK / (4 * NTH)], const float *ptr, int ld, int nx, int vec,
                                    int x0, int k0, int K, int tid) {
    static_assert(!COH || FAST, "coherent register path: 16-byte-legal operands only");
    constexpr int NV = TX * BK / (4 * NTH);
    constexpr int RC = (L == KM) ? TX / 4 : BK / 4;      // float4 per tile row
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int f = tid + n * NTH;
        const int row = f / RC, c4 = f % RC;
        const int k = (L == KM) ? k0 + row : k0 + c4 * 4;
        const int x = (L == KM) ? x0 + c4 * 4 : x0 + row;
        if (FAST) {
            const int kc = (L == KM) ? min(k, K - 1) : min(k, K - 4);
            const int xc = (L == KM) ? min(x, nx - 4) : min(x, nx - 1);
            const float *pc = (L == KM) ? ptr + (size_t)kc * ld + xc : ptr + (size_t)xc * ld + kc;
            reg[n] = COH ? load4_coh(pc) : *reinterpret_cast<const float4 *>(pc);
        } else if (L == KM) {
            reg[n] = load4_guard(ptr + (size_t)k * ld + x, k < K, x, nx, vec);
        } else {
            reg[n] = load4_guard(ptr + (size_t)x * ld + k, x < nx, k, K, vec);
        }
    }
}

// registers -> LDS image.  kz = K - k0: rows / columns of this chunk at k >= K are zeroed.
template <int L, int TX, int BK, int NTH, int S, bool ZF = true>
__device__ __forceinline__ void r2s(const float4 (&reg)[TX * BK / (4 * NTH)], float *s, int tid, int kz) {
    constexpr int NV = TX * BK / (4 * NTH);
    constexpr int RC = (L == KM) ? TX / 4 : BK / 4;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int f = tid + n * NTH;
        const int row = f / RC, c4 = f % RC;
        float4 v = reg[n];
        if (!ZF) {
        } else if (L == KM) {
            if (row >= kz) v = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            const int k = c4 * 4;
            if (k     >= kz) v.x = 0.f;
            if (k + 1 >= kz) v.y = 0.f;
            if (k + 2 >= kz) v.z = 0.f;
            if (k + 3 >= kz) v.w = 0.f;
        }
        *reinterpret_cast<float4 *>(s + img_off<L, TX, BK, S>(row, c4)) = v;
    }
}

// MFMA operand fragments of one BK chunk for this wave
template <class G> struct Frags {
    float p[G::BK / 4][G::MI];     // p[4m+j][t] = P[k = 16m + 4g + j][i = base + MI*l15 + t]
    float q[G::BK / 4][G::NJ];     // q[4m+j][n] = Q[j' = lane_j(l15, n)][k = 16m + 4g + j]
};

template <int QL, class G, int ABL = 0, int PL = KM>
__device__ __forceinline__ void read_frags(Frags<G> &f, const float *sP, const float *sQ, int wi, int wj, int lane) {
    static_assert(PL == KM || G::MI == 1, "x-major P needs MI == 1 (no interleaved sub-tiles)");
    constexpr int BK = G::BK;
    const int g = lane >> 4, l15 = lane & 15;
    if (BM_ABL(3)) {
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) { f.p[kk][0] = 1.f; f.p[kk][G::MI - 1] = 2.f; f.q[kk][0] = 1.f; f.q[kk][G::NJ - 1] = 1.f; }
        return;
    }
    // ---- P
    if (PL == XM) {
        const int row = wi * 16 + l15;
#pragma unroll
        for (int m = 0; m < BK / 16; ++m) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(sP + img_off<XM, 0, BK, 0>(row, 4 * m + g));
#pragma unroll
            for (int j = 0; j < 4; ++j) f.p[4 * m + j][0] = v[j];
        }
    } else {
        const int col = wi * (16 * G::MI) + G::MI * l15;          // MI == 2: even, both floats in one chunk
#pragma unroll
        for (int m = 0; m < BK / 16; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 16 * m + 4 * g + j;
                const float *a = sP + row * G::TI + (((col >> 2) ^ fk<G::SP>(row)) << 2) + (col & 3);
                if (G::MI == 2) {
                    const f32x2v t = *reinterpret_cast<const f32x2v *>(a);
                    f.p[4 * m + j][0] = t[0]; f.p[4 * m + j][G::MI - 1] = t[1];
                } else {
                    f.p[4 * m + j][0] = *a;
                }
            }
    }
    // ---- Q
    if (QL == XM) {
#pragma unroll
        for (int n = 0; n < G::NJ; ++n) {
            const int row = wj * (16 * G::NJ) + 16 * n + l15;
#pragma unroll
            for (int m = 0; m < BK / 16; ++m) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(sQ + img_off<XM, 0, BK, 0>(row, 4 * m + g));
#pragma unroll
                for (int j = 0; j < 4; ++j) f.q[4 * m + j][n] = v[j];
            }
        }
    } else {
        const int col = wj * (16 * G::NJ) + G::NJ * l15;          // NJ == 2: sub-tile n holds j = base + 2*l15 + n
#pragma unroll
        for (int m = 0; m < BK / 16; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 16 * m + 4 * g + j;
                const float *a = sQ + row * G::TJ + (((col >> 2) ^ fk<G::SQ>(row)) << 2) + (col & 3);
                if (G::NJ == 2) {
                    const f32x2v t = *reinterpret_cast<const f32x2v *>(a);
                    f.q[4 * m + j][0] = t[0]; f.q[4 * m + j][G::NJ - 1] = t[1];
                } else {
                    f.q[4 * m + j][0] = *a;
                }
            }
    }
}

// acc[t][n] += P-frag(t) x Q-frag(n), every k block of the chunk
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

// the same for the first `nq` blocks of 16 k only (last chunk of a contraction whose K tail is shorter than BK:
// the zero-filled remainder would only add fma(0, 0, acc))
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

// one register set = one BK chunk of both operand tiles (register path)
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

// branch-free wave-uniform selects
__device__ __forceinline__ int sel_i(int a, int b, int m) { return a ^ ((a ^ b) & m); }
__device__ __forceinline__ const float *sel_p(const float *a, const float *b, int m) {
    const uintptr_t ua = (uintptr_t)a, ub = (uintptr_t)b;
    return (const float *)(ua ^ ((ua ^ ub) & (uintptr_t)(intptr_t)m));
}

// ---- LDS-DMA plan: per-lane byte offsets (relative to the chunk base of the operand) of the 16-byte
// global chunk this lane moves in each of the wave's pieces; piece n of wave w is piece p = w + n*NW of the
// tile image (1 KiB = 64 consecutive 16-byte slots: lane l fills slot l of the piece).
template <class G, int DW = G::NW> struct DmaPlan {
    static constexpr int NPP = G::P_BUF / (256 * DW), NPQ = G::Q_BUF / (256 * DW);   // pieces per DMA wave and chunk
    uint32_t p[NPP], q[NPQ];
};

template <int L, int TX, int BK, int NW, int NP, int S>
__device__ __forceinline__ void plan_offsets(uint32_t (&off)[NP], int ld, int nx, int x0, int wave, int lane) {
    constexpr int RC = (L == KM) ? TX / 4 : BK / 4;      // 16-byte slots per image row
    constexpr int R = 64 / RC;                           // image rows per piece
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        const int pc = wave + n * NW;
        const int row = pc * R + lane / RC, slot = lane % RC;
        if (L == KM) {
            const int c4 = slot ^ fk<S>(row);
            off[n] = (uint32_t)(row * ld + min(x0 + c4 * 4, nx - 4)) * 4u;
        } else {
            const int c4 = slot ^ fx<BK>(row);
            off[n] = (uint32_t)(min(x0 + row, nx - 1) * ld + c4 * 4) * 4u;
        }
    }
}

template <int QL, class G, int PL = KM, int DW = G::NW>
__device__ __forceinline__ void make_plan(DmaPlan<G, DW> &pl, const Operand &P, const Operand &Q, int i0, int j0, int wave, int lane) {
    plan_offsets<PL, G::TI, G::BK, DW, DmaPlan<G, DW>::NPP, G::SP>(pl.p, P.ld, P.nx, i0, wave, lane);
    plan_offsets<QL, G::TJ, G::BK, DW, DmaPlan<G, DW>::NPQ, G::SQ>(pl.q, Q.ld, Q.nx, j0, wave, lane);
}

// One LDS-DMA wave instruction: 64 lanes x 16 bytes from each lane's `src` to lds_dst + 16 * lane.
// Inline asm, not __builtin_amdgcn_global_load_lds: with the builtin hipcc (ROCm 7.2) treats every later
// ds_read as a possible reader of the DMA's destination and puts `s_waitcnt vmcnt(0)` in front of it, i.e.
// it drains the DMA queue in the very step that filled it.  The asm form is invisible to that bookkeeping:
// completion is counted by hand (BM_WAIT_VM + barrier before the first read of a slot; cdna_hip_programming.md
// "What hipcc does not do").  M0 (the DMA's LDS base) is written in the same statement that uses it.
__device__ __forceinline__ void dma16(const char *src, float *lds_dst) {
    unsigned keep;
    const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds_dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
}

// The same with a SCALAR global base and a 32-bit per-lane byte offset (`global_load_lds_dwordx4 v, s[..]`), and
// the LDS destination as a wave-uniform BYTE address: no vector ALU instruction per piece.  That matters more than
// it looks: the MFMA stream of a wave is one dependent chain, and tools/ubench_step.hip measures ~13 cycles of
// matrix-pipe time lost per VALU instruction slipped between two MFMAs (16 MFMAs + 12 LDS reads + barrier: 1055
// cycles per step; one v_add_u32 behind every MFMA: 1470), while scalar instructions are nearly free (4 per MFMA:
// 1098).  The steady steps below are therefore written so that hipcc needs no VALU for addresses at all.
__device__ __forceinline__ void dma16s(const void *sbase, uint32_t voff, unsigned lds_byte) {
    // M0 is written and NOT restored (two scalar moves per DMA cost 0.6 us per update): nothing else in these kernels
    // reads it - gfx9+ DS instructions do not, and hipcc emits no LDS-direct / movrel / GWS / sendmsg here (the
    // generated ISA is checked for stray M0 uses by tests/test_host_logic.py).  hipcc refuses M0 in a clobber list
    // (reserved register), so it is not declared.
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0"
                 :: "s"(sbase), "v"(voff), "s"(lds_byte) : "memory");
}
// the same with sc1: the load bypasses the CU's vector L1 and is served by the XCD's L2 (operands written by another
// workgroup of the SAME launch on the same XCD, bm_chain.h)
__device__ __forceinline__ void dma16s_coh(const void *sbase, uint32_t voff, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 sc1"
                 :: "s"(sbase), "v"(voff), "s"(lds_byte) : "memory");
}
// chunk `kc` of ONE segment (full chunk): scalar chunk bases pb / qb, plan of that segment, LDS byte addresses of
// this wave's first piece in the destination slot images
template <class G, int DW, bool QSC = false, bool QHALF = false>
__device__ __forceinline__ void dma_chunk_s(const DmaPlan<G, DW> &pl, const char *pb, const char *qb, unsigned ldsP, unsigned ldsQ) {
#pragma unroll
    for (int n = 0; n < DmaPlan<G, DW>::NPP; ++n) dma16s(pb, pl.p[n], ldsP + (unsigned)(n * DW * 1024));
#pragma unroll
    for (int n = 0; n < DmaPlan<G, DW>::NPQ; ++n) {
        if (QHALF && (n & 1)) continue;
        if (QSC) dma16s_coh(qb, pl.q[n], ldsQ + (unsigned)(n * DW * 1024));
        else     dma16s(qb, pl.q[n], ldsQ + (unsigned)(n * DW * 1024));
    }
}

// chunk c (a FULL chunk of its segment) -> LDS slot images sP / sQ, all pieces of this wave
template <int QL, class G, bool SEG2, int PL = KM, int DW = G::NW, bool QHALF = false>
__device__ __forceinline__ void dma_chunk(const KRange &kr, const DmaPlan<G, DW> &pl1, const DmaPlan<G, DW> &pl2, int nch1, int c,
                                          float *sP, float *sQ, int wave) {
    constexpr int BK = G::BK;
    const int m = SEG2 ? -(int)(c >= nch1) : 0;           // all-ones in segment 2 (wave-uniform)
    const int kc = c - (nch1 & m);
    const int ldp = SEG2 ? sel_i(kr.P1.ld, kr.P2.ld, m) : kr.P1.ld;
    const int ldq = SEG2 ? sel_i(kr.Q1.ld, kr.Q2.ld, m) : kr.Q1.ld;
    const char *pb = (const char *)((SEG2 ? sel_p(kr.P1.ptr, kr.P2.ptr, m) : kr.P1.ptr) +
                                    ((PL == KM) ? (size_t)kc * BK * ldp : (size_t)kc * BK));
    const char *qb = (const char *)((SEG2 ? sel_p(kr.Q1.ptr, kr.Q2.ptr, m) : kr.Q1.ptr) +
                                    ((QL == KM) ? (size_t)kc * BK * ldq : (size_t)kc * BK));
#pragma unroll
    for (int n = 0; n < DmaPlan<G, DW>::NPP; ++n) {
        const uint32_t o1 = pl1.p[n], o2 = SEG2 ? pl2.p[n] : 0u;
        const uint32_t o = SEG2 ? (o1 ^ ((o1 ^ o2) & (uint32_t)m)) : o1;
        dma16(pb + o, sP + (wave + n * DW) * 256);
    }
#pragma unroll
    for (int n = 0; n < DmaPlan<G, DW>::NPQ; ++n) {
        if (QHALF && (n & 1)) continue;
        const uint32_t o1 = pl1.q[n], o2 = SEG2 ? pl2.q[n] : 0u;
        const uint32_t o = SEG2 ? (o1 ^ ((o1 ^ o2) & (uint32_t)m)) : o1;
        dma16(qb + o, sQ + (wave + n * DW) * 256);
    }
}

// register path: load chunk c (any chunk) / store it into the slot images with the K-tail zero fill
template <int QL, class G, bool FAST, bool SEG2, int PL = KM, bool COHQ = false>
__device__ __forceinline__ void load_chunk(ChunkRegs<G> &r, const KRange &kr, int nch1, int i0, int j0, int c, int tid) {
    constexpr int BK = G::BK;
    if (!SEG2) {
        g2r<PL, G::TI, BK, G::NT, FAST>(r.p, kr.P1.ptr, kr.P1.ld, kr.P1.nx, kr.P1.vec, i0, c * BK, kr.K1, tid);
        g2r<QL, G::TJ, BK, G::NT, FAST, COHQ>(r.q, kr.Q1.ptr, kr.Q1.ld, kr.Q1.nx, kr.Q1.vec, j0, c * BK, kr.K1, tid);
    } else {
        const int m = -(int)((c >= nch1) & (kr.K2 > 0));   // all-ones in segment 2 (wave-uniform)
        const int kc = c - (nch1 & m);
        const int K = sel_i(kr.K1, kr.K2, m);
        g2r<PL, G::TI, BK, G::NT, FAST>(r.p, sel_p(kr.P1.ptr, kr.P2.ptr, m), sel_i(kr.P1.ld, kr.P2.ld, m),
                                        sel_i(kr.P1.nx, kr.P2.nx, m), sel_i(kr.P1.vec, kr.P2.vec, m), i0, kc * BK, K, tid);
        g2r<QL, G::TJ, BK, G::NT, FAST, COHQ>(r.q, sel_p(kr.Q1.ptr, kr.Q2.ptr, m), sel_i(kr.Q1.ld, kr.Q2.ld, m),
                                              sel_i(kr.Q1.nx, kr.Q2.nx, m), sel_i(kr.Q1.vec, kr.Q2.vec, m), j0, kc * BK, K, tid);
    }
}

template <int QL, class G, bool SEG2, int PL = KM>
__device__ __forceinline__ void store_chunk(const ChunkRegs<G> &r, const KRange &kr, int nch1, int c,
                                            float *sP, float *sQ, int tid) {
    constexpr int BK = G::BK;
    const int m = SEG2 ? -(int)((c >= nch1) & (kr.K2 > 0)) : 0;
    const int kz = sel_i(kr.K1, kr.K2, m) - (c - (nch1 & m)) * BK;
    r2s<PL, G::TI, BK, G::NT, G::SP>(r.p, sP, tid, kz);
    r2s<QL, G::TJ, BK, G::NT, G::SQ>(r.q, sQ, tid, kz);
}

// ---- register staging of FULL chunks ("REG" steady steps; the alternative to LDS-DMA, chosen per shape by
// measurement: a DMA wave instruction holds the issuing wave ~75 cycles, which the 4-wave outer-product kernel
// - one wave per SIMD, 8 pieces per wave and step - cannot hide, while global_load_dwordx4 + ds_write_b128 cost it
// less issue time).  Per-thread byte offsets are computed once per kernel on top of a wave-uniform chunk base.
template <class G> struct RegPlan {
    uint32_t p[G::NVP], q[G::NVQ];
};
template <int L, int TX, int BK, int NTH>
__device__ __forceinline__ void reg_offsets(uint32_t (&off)[TX * BK / (4 * NTH)], int ld, int nx, int x0, int tid) {
    constexpr int NV = TX * BK / (4 * NTH);
    constexpr int RC = (L == KM) ? TX / 4 : BK / 4;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int f = tid + n * NTH;
        const int row = f / RC, c4 = f % RC;
        off[n] = (L == KM) ? (uint32_t)(row * ld + min(x0 + c4 * 4, nx - 4)) * 4u
                           : (uint32_t)(min(x0 + row, nx - 1) * ld + c4 * 4) * 4u;
    }
}
template <int QL, class G, int PL = KM>
__device__ __forceinline__ void make_reg_plan(RegPlan<G> &pl, const Operand &P, const Operand &Q, int i0, int j0, int tid) {
    reg_offsets<PL, G::TI, G::BK, G::NT>(pl.p, P.ld, P.nx, i0, tid);
    reg_offsets<QL, G::TJ, G::BK, G::NT>(pl.q, Q.ld, Q.nx, j0, tid);
}
// chunk c (a FULL chunk of its segment) -> registers
template <int QL, class G, bool SEG2, int PL = KM>
__device__ __forceinline__ void load_chunk_slim(ChunkRegs<G> &r, const KRange &kr, const RegPlan<G> &pl1,
                                                const RegPlan<G> &pl2, int nch1, int c) {
    constexpr int BK = G::BK;
    const int m = SEG2 ? -(int)(c >= nch1) : 0;           // all-ones in segment 2 (wave-uniform)
    const int kc = c - (nch1 & m);
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
    r2s<PL, G::TI, G::BK, G::NT, G::SP, false>(r.p, sP, tid, 0);
    r2s<QL, G::TJ, G::BK, G::NT, G::SQ, false>(r.q, sQ, tid, 0);
}

// STG_DMAH ("half"): in the 8-wave geometries only waves 0 .. 3 - one per SIMD - issue the DMA instructions (twice
// as many each); their SIMD partners 4 .. 7 go straight to the MFMAs, so the time a DMA instruction holds its
// wave is covered by the partner's matrix work instead of stalling both waves of the SIMD at the same moment.
enum : int { STG_DMA = 0, STG_REG = 1, STG_DMAH = 2 };

// counted waits / raw barrier (a __syncthreads() would insert vmcnt(0) and drain the DMA queue)
#define BM_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// wait until at most n (wave-uniform, 0 .. MAXC) chunks' worth of this wave's DMA instructions are in flight
template <int NPW, int MAXC>
__device__ __forceinline__ void wait_vm_chunks(int n) {
    static_assert(MAXC * NPW <= 63, "vmcnt range");
    if (MAXC >= 4 && n >= 4)      BM_WAIT_VM(4 * NPW <= 63 ? 4 * NPW : 0);
    else if (MAXC >= 3 && n == 3) BM_WAIT_VM(3 * NPW <= 63 ? 3 * NPW : 0);
    else if (MAXC >= 2 && n == 2) BM_WAIT_VM(2 * NPW <= 63 ? 2 * NPW : 0);
    else if (n == 1)              BM_WAIT_VM(NPW);
    else                          BM_WAIT_VM(0);
}
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS reads / writes have completed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// acc += sum_k P[k][i] * Q[j][k] over the K range in canonical order (header).
//
// Pipeline: chunk c lives in LDS slot c % 4.  Step c (between barriers B(c-1) and B(c)):
//   * starts chunk c+3 by LDS-DMA into slot (c+3) % 4 (last read for F(c-1) in step c-2),
//   * runs the MFMAs of chunk c on fragments F(c) that are ALREADY in registers,
//   * reads the fragments F(c+1) from slot (c+1) % 4,
//   * waits until its own DMA pieces of chunk c+2 have landed (vmcnt(pieces of chunk c+3)), then B(c).
// A chunk that touches the end of a segment (and every chunk when !FAST) goes through registers instead:
// loaded at the start of step c-2, written to its slot at the end of that step.  The leading run of steps
// whose DMA chunk and whose next-but-one chunk are both DMA chunks is straight-line code ("steady" steps:
// sched_group_barrier pins the MFMA / DMA / LDS-read interleave); the rest takes the general step.
// `side.fill()` (the lane's Philox blocks in act_kernel) runs while the first loads are in flight.
template <int QL, class G, bool FAST, bool SEG2, int ABL = 0, int PL = KM, int STG = STG_DMA, class Side = NoSide>
__device__ __forceinline__ void mainloop(f32x4 (&acc)[G::MI][G::NJ], const KRange &kr, int i0, int j0, float *smem,
                                         Side &side, long long *stamps = nullptr) {
#ifdef BM_PROBE
#define BM_MSTAMP(n) do { if (stamps && threadIdx.x == 0) stamps[n] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define BM_MSTAMP(n) do {} while (0)
#endif
    constexpr int BK = G::BK, P_BUF = G::P_BUF, Q_BUF = G::Q_BUF, NBUF = G::NBUF, PF = G::PF;
    constexpr int DW = (STG == STG_DMAH && G::NW == 8) ? 4 : G::NW;       // waves that issue DMA instructions
    constexpr int NPW = DmaPlan<G, DW>::NPP + (BM_ABL(9) ? (DmaPlan<G, DW>::NPQ + 1) / 2 : DmaPlan<G, DW>::NPQ);   // DMA instructions per DMA wave and chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform (SGPR): LDS piece bases stay scalar
    const int wi = w % G::WI, wj = w / G::WI;
    float *sP = smem, *sQ = smem + NBUF * P_BUF;
    const int nch1 = (kr.K1 + BK - 1) / BK;
    const int nch = nch1 + (SEG2 ? (kr.K2 + BK - 1) / BK : 0);
    const int nfull1 = kr.K1 / BK, nfull2 = SEG2 ? kr.K2 / BK : 0;
    // can chunk c be moved by DMA?  (a full chunk of its segment, 16-byte-legal operands)
    auto is_full = [&](int c) -> bool {               // a full chunk of its segment
        return (c < nfull1) | (SEG2 && c >= nch1 && c - nch1 < nfull2);
    };
    auto is_dma = [&](int c) -> bool {
        if (!FAST || STG == STG_REG || BM_ABL(6)) return false;
        return is_full(c);
    };
    ChunkRegs<G> gr;             // register-path staging (one chunk)
    Frags<G> fa, fb;
    DmaPlan<G, DW> pl1, pl2;
    RegPlan<G> rp1, rp2;
    const bool dmaw = (DW == G::NW) || w < DW;       // wave-uniform
    ChunkRegs<G> g0, g1;         // REG staging: the two register sets of the steady steps (named, never arrays)
    if (FAST && STG != STG_REG && dmaw) {
        make_plan<QL, G, PL, DW>(pl1, kr.P1, kr.Q1, i0, j0, w, lane);
        if (SEG2) make_plan<QL, G, PL, DW>(pl2, kr.P2, kr.Q2, i0, j0, w, lane);
    }
    // REG staging: leading run of steps c whose chunks c+2 (stored) and c+4 (loaded) are full chunks; chunks
    // 0 .. 3 are then full as well and go through the same slim path in the fill
    int n_reg = 0;
    if (FAST && STG == STG_REG && !BM_ABL(6) && !BM_ABL(0)) {
        while (n_reg + 4 < nch && is_full(n_reg + 4) && is_full(n_reg + 2)) ++n_reg;
        n_reg &= ~1;
        if (n_reg > 0 && !(is_full(0) && is_full(1) && is_full(2) && is_full(3))) n_reg = 0;
        if (n_reg > 0) {
            make_reg_plan<QL, G, PL>(rp1, kr.P1, kr.Q1, i0, j0, tid);
            if (SEG2) make_reg_plan<QL, G, PL>(rp2, kr.P2, kr.Q2, i0, j0, tid);
        }
    }
    BM_MSTAMP(0);
    if (n_reg > 0) {
        // ---- pipeline fill, REG staging: the four chunk loads go out back to back (ONE memory round trip);
        // chunks 0 / 1 pass through two prologue-only sets, chunks 2 / 3 land in the loop's sets
        ChunkRegs<G> ga, gb;
        load_chunk_slim<QL, G, SEG2, PL>(ga, kr, rp1, rp2, nch1, 0);
        load_chunk_slim<QL, G, SEG2, PL>(gb, kr, rp1, rp2, nch1, 1);
        load_chunk_slim<QL, G, SEG2, PL>(g0, kr, rp1, rp2, nch1, 2);
        load_chunk_slim<QL, G, SEG2, PL>(g1, kr, rp1, rp2, nch1, 3);
        __builtin_amdgcn_sched_barrier(0);
        side.fill();
        __builtin_amdgcn_sched_barrier(0);
        store_chunk_slim<QL, G, PL>(ga, sP, sQ, tid);
        store_chunk_slim<QL, G, PL>(gb, sP + P_BUF, sQ + Q_BUF, tid);
    } else if constexpr (Side::kSplitFill) {
        // ---- pipeline fill of a chained launch (bm_chain.h; the host guarantees FAST, DMA staging, one segment and
        // K1 >= PF * BK: chunks 0 .. PF-1 are full DMA chunks).  The P pieces (weights: constant during the launch) go
        // out first, then the wave waits for the producers of its Q rows, then the Q pieces.  The vector-memory queue
        // completes in order: with only the Q pieces of chunk PF-1 .. 2 still in flight, chunks 0 and 1 are complete.
        static_assert(FAST && STG == STG_DMA && DW == G::NW, "chained fill");
        const unsigned l0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
        const unsigned lP = l0 + (unsigned)w * 1024u, lQ = l0 + (unsigned)(NBUF * P_BUF * 4) + (unsigned)w * 1024u;
#pragma unroll
        for (int c = 0; c < PF; ++c) {
            const char *pb = (const char *)kr.P1.ptr + (size_t)c * ((PL == KM) ? (size_t)BK * kr.P1.ld * 4 : (size_t)BK * 4);
#pragma unroll
            for (int n = 0; n < DmaPlan<G, DW>::NPP; ++n) dma16s(pb, pl1.p[n], lP + (unsigned)(c * P_BUF * 4 + n * DW * 1024));
        }
        __builtin_amdgcn_sched_barrier(0);
        side.fill();
        __builtin_amdgcn_sched_barrier(0);
        side.wait_inputs();
#pragma unroll
        for (int c = 0; c < PF; ++c) {
            const char *qb = (const char *)kr.Q1.ptr + (size_t)c * ((QL == KM) ? (size_t)BK * kr.Q1.ld * 4 : (size_t)BK * 4);
#pragma unroll
            for (int n = 0; n < DmaPlan<G, DW>::NPQ; ++n) dma16s_coh(qb, pl1.q[n], lQ + (unsigned)(c * Q_BUF * 4 + n * DW * 1024));
        }
        constexpr int NQ_LEFT = (PF - 2) * DmaPlan<G, DW>::NPQ;
        BM_WAIT_VM(NQ_LEFT);
    } else {
        // ---- pipeline fill: chunks 0 .. PF-1
#pragma unroll
        for (int c = 0; c < PF; ++c)
            if (c < nch && is_dma(c) && dmaw && !BM_ABL(0))
                dma_chunk<QL, G, SEG2, PL, DW, (BM_ABL(9) != 0)>(kr, pl1, pl2, nch1, c, sP + (c % NBUF) * P_BUF, sQ + (c % NBUF) * Q_BUF, w);
        __builtin_amdgcn_sched_barrier(0);
        side.fill();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 2; ++c)                      // register-path chunks among the first two: synchronously
            if (c < nch && !is_dma(c) && !BM_ABL(0)) {
                load_chunk<QL, G, FAST, SEG2, PL>(gr, kr, nch1, i0, j0, c, tid);
                store_chunk<QL, G, SEG2, PL>(gr, kr, nch1, c, sP + (c % NBUF) * P_BUF, sQ + (c % NBUF) * Q_BUF, tid);
            }
        // chunks 0 and 1 complete (the DMAs of chunks 2 .. PF-1, if any, may stay in flight)
        {
            int n_after = 0;
#pragma unroll
            for (int c = 2; c < PF; ++c) n_after += (c < nch && is_dma(c) && !BM_ABL(0)) ? 1 : 0;
            wait_vm_chunks<NPW, PF - 2>(n_after);
        }
    }
    BM_MSTAMP(1);
    if (!BM_ABL(5)) wg_barrier();
    if constexpr (Side::kCanAbort) {
        side.post_fill();
        if (side.aborted) { BM_WAIT_VM(0); return; }     // (an LDS-DMA piece must not land in the LDS of a finished workgroup)
    }
    read_frags<QL, G, ABL, PL>(fa, sP, sQ, wi, wj, lane);
    BM_MSTAMP(2);
    // blocks of 16 k that the last chunk really holds
    const int klast = (SEG2 && kr.K2 > 0) ? kr.K2 - (nch - nch1 - 1) * BK : kr.K1 - (nch1 - 1) * BK;
    const int nq_last = (klast + 15) / 16;
    // Issue order of a steady step: the DMA instructions behind the first MFMAs (longest latency), then one LDS
    // read behind every further MFMA.   masks: 0x008 MFMA, 0x100 DS read, 0x020 VMEM read
    constexpr int NM = (BK / 4) * G::MI * G::NJ;
    constexpr int NRP = (PL == XM) ? BK / 16 : BK / 4;
    constexpr int NRQ = (QL == XM) ? (BK / 16) * G::NJ : BK / 4;
    constexpr int NR = NRP + NRQ;
#define BM_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0);
#ifndef BM_SCHED_VARIANT
#define BM_SCHED_VARIANT 0
#endif
#if BM_SCHED_VARIANT == 0
#define BM_SCHED_STEP                                                                             \
    if (DW == G::NW) { _Pragma("unroll") for (int s_ = 0; s_ < NPW; ++s_) { BM_SG(0x008, 1) BM_SG(0x020, 1) } } \
    _Pragma("unroll") for (int s_ = 0; s_ < NR; ++s_) { BM_SG(0x008, 1) BM_SG(0x100, 1) }         \
    BM_SG(0x008, NM)
#elif BM_SCHED_VARIANT == 1    /* reads first, DMA behind them */
#define BM_SCHED_STEP                                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < NR; ++s_) { BM_SG(0x008, 1) BM_SG(0x100, 1) }         \
    _Pragma("unroll") for (int s_ = 0; s_ < NPW; ++s_) { BM_SG(0x008, 1) BM_SG(0x020, 1) }        \
    BM_SG(0x008, NM)
#else                          /* no pinning: leave the order to hipcc */
#define BM_SCHED_STEP
#endif
    static_assert(NBUF == 4, "the steps below are written for a 4-slot ring");
    int cc = 0;
    // EVERY step works on COMPILE-TIME ring slots: the loops are unrolled by four and step cc uses slot S = cc % 4,
    // so each LDS address is a loop-invariant base plus an immediate (hipcc hoists them), and the DMA takes a
    // scalar chunk base.  The steady steps contain no vector ALU instruction at all (see dma16s for why it matters).
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
    const unsigned ldsPw = lds0 + (unsigned)w * 1024u, ldsQw = lds0 + (unsigned)(NBUF * P_BUF * 4) + (unsigned)w * 1024u;
    // scalar base of chunk c of its segment, and the plan of that segment (swapped in once, under a scalar branch)
    bool in2 = false;                                 // pl1 holds the segment-2 plan from then on
    auto chunk_bases = [&](int c, const char *&pb, const char *&qb) {
        const bool s2 = SEG2 && c >= nch1;
        const Operand &P = s2 ? kr.P2 : kr.P1, &Q = s2 ? kr.Q2 : kr.Q1;
        const int kc = s2 ? c - nch1 : c;
        pb = (const char *)P.ptr + (size_t)kc * ((PL == KM) ? (size_t)BK * P.ld * 4 : (size_t)BK * 4);
        qb = (const char *)Q.ptr + (size_t)kc * ((QL == KM) ? (size_t)BK * Q.ld * 4 : (size_t)BK * 4);
    };
#define BM_DMA_AT(CD, SD)                                                                         \
    {                                                                                             \
        if (SEG2 && !in2 && (CD) >= nch1) { in2 = true; pl1 = pl2; }                              \
        const char *pb_, *qb_;                                                                    \
        chunk_bases((CD), pb_, qb_);                                                              \
        if (dmaw) dma_chunk_s<G, DW, Side::kCohQ, (BM_ABL(9) != 0)>(pl1, pb_, qb_, ldsPw + (unsigned)((SD) * P_BUF * 4), \
                                     ldsQw + (unsigned)((SD) * Q_BUF * 4));                       \
    }
    // steady step: chunk cc+3 by DMA, chunk cc+2 a DMA chunk as well (nothing passes through registers)
#define BM_STEP_Q(FC, FN, S)                                                                      \
    {                                                                                             \
        BM_DMA_AT(cc + PF, ((S) + PF) % NBUF)                                                     \
        read_frags<QL, G, ABL, PL>(FN, sP + (((S) + 1) % NBUF) * P_BUF, sQ + (((S) + 1) % NBUF) * Q_BUF, wi, wj, lane); \
        mfma_frags<G, ABL>(acc, FC);                                                              \
        BM_SCHED_STEP                                                                             \
        if (!BM_ABL(8)) BM_WAIT_VM((PF - 2) * NPW);                                               \
        if (!BM_ABL(5)) wg_barrier();                                                             \
        ++cc;                                                                                     \
    }
    // general step: FC = fragments of chunk cc, FN <- fragments of chunk cc+1; chunk cc+3 by DMA if it is a DMA
    // chunk, chunk cc+2 through registers if it is not
#define BM_STEP_T(FC, FN, S1, S2, S3)      /* S1 / S2 / S3: ring slots of chunks cc+1 / cc+2 / cc+3 */ \
    {                                                                                             \
        const int cd = cc + PF, cr = cc + 2;                                                      \
        const bool do_dma = cd < nch && is_dma(cd) && !BM_ABL(0);                                 \
        const bool do_reg = cr < nch && !is_dma(cr) && !BM_ABL(0);                                \
        if (cc == last - 1) side.drain();      /* step nch-2: every operand load has landed (waited in step nch-3) */ \
        if (do_reg) load_chunk<QL, G, FAST, SEG2, PL, Side::kCohQ>(gr, kr, nch1, i0, j0, cr, tid); \
        if (do_dma) BM_DMA_AT(cd, S3)                                                             \
        read_frags<QL, G, ABL, PL>(FN, sP + (S1) * P_BUF, sQ + (S1) * Q_BUF, wi, wj, lane);       \
        mfma_frags<G, ABL>(acc, FC);                                                              \
        _Pragma("unroll") for (int s_ = 0; s_ < NR; ++s_) { BM_SG(0x008, 1) BM_SG(0x100, 1) }     \
        BM_SG(0x008, NM)                                                                          \
        if (do_reg) {                                                                             \
            BM_WAIT_VM(0);                                                                        \
            store_chunk<QL, G, SEG2, PL>(gr, kr, nch1, cr, sP + (S2) * P_BUF, sQ + (S2) * Q_BUF, tid); \
        } else if (cr < nch) {                                                                    \
            /* chunk cc+2 must have landed: the DMAs of chunks cc+3 .. cc+PF may stay in flight */ \
            int n_after = 0;                                                                      \
            _Pragma("unroll") for (int c_ = 3; c_ <= PF; ++c_) n_after += (cc + c_ < nch && is_dma(cc + c_) && !BM_ABL(0)) ? 1 : 0; \
            wait_vm_chunks<NPW, PF - 2>(n_after);                                                 \
        }                                                                                         \
        if (!BM_ABL(5)) wg_barrier();                                                             \
        ++cc;                                                                                     \
    }
#define BM_STEP_TC(FC, FN, S) BM_STEP_T(FC, FN, ((S) + 1) % NBUF, ((S) + 2) % NBUF, ((S) + 3) % NBUF)
#define BM_STEP_TR(FC, FN)    BM_STEP_T(FC, FN, (cc + 1) % NBUF, (cc + 2) % NBUF, (cc + 3) % NBUF)
    // REG steady step: chunk cc+2 (register set G_) -> its slot, chunk cc+4 -> the same set (RELOAD), fragments of
    // chunk cc+1, MFMAs of chunk cc.   masks: 0x008 MFMA, 0x100 DS read, 0x200 DS write, 0x020 VMEM read
    constexpr int NWR = G::NVP + G::NVQ;              // DS writes / global loads per thread and chunk
#ifndef BM_REG_SCHED_VARIANT
#define BM_REG_SCHED_VARIANT 0
#endif
#if BM_REG_SCHED_VARIANT == 0      /* writes, loads, reads: each behind one MFMA */
#define BM_REG_SCHED(RELOAD)                                                                      \
        _Pragma("unroll") for (int s_ = 0; s_ < NWR; ++s_) { BM_SG(0x008, 1) BM_SG(0x200, 1) }    \
        if (RELOAD) { _Pragma("unroll") for (int s_ = 0; s_ < NWR; ++s_) { BM_SG(0x008, 1) BM_SG(0x020, 1) } } \
        _Pragma("unroll") for (int s_ = 0; s_ < NR; ++s_) { BM_SG(0x008, 1) BM_SG(0x100, 1) }     \
        BM_SG(0x008, NM)
#elif BM_REG_SCHED_VARIANT == 1    /* round 1's order: reads (2 MFMAs apart when there are >= 64), writes, loads */
#define BM_REG_SCHED(RELOAD)                                                                      \
        _Pragma("unroll") for (int s_ = 0; s_ < NR; ++s_) { BM_SG(0x008, (NM >= 64 ? 2 : 1)) BM_SG(0x100, 2) } \
        _Pragma("unroll") for (int s_ = 0; s_ < NWR; ++s_) { BM_SG(0x008, (NM >= 32 ? 2 : 1)) BM_SG(0x200, 1) } \
        if (RELOAD) { _Pragma("unroll") for (int s_ = 0; s_ < NWR; ++s_) { BM_SG(0x008, 1) BM_SG(0x020, 1) } } \
        BM_SG(0x008, NM)
#else                               /* no pinning */
#define BM_REG_SCHED(RELOAD)
#endif
#define BM_STEP_REG(FC, FN, G_, RELOAD, S)                                                        \
    {                                                                                             \
        store_chunk_slim<QL, G, PL>(G_, sP + (((S) + 2) % NBUF) * P_BUF, sQ + (((S) + 2) % NBUF) * Q_BUF, tid); \
        if (RELOAD) load_chunk_slim<QL, G, SEG2, PL>(G_, kr, rp1, rp2, nch1, cc + 4);             \
        read_frags<QL, G, ABL, PL>(FN, sP + (((S) + 1) % NBUF) * P_BUF, sQ + (((S) + 1) % NBUF) * Q_BUF, wi, wj, lane); \
        mfma_frags<G, ABL>(acc, FC);                                                              \
        BM_REG_SCHED(RELOAD)                                                                      \
        if (!BM_ABL(5)) wg_barrier();                                                             \
        ++cc;                                                                                     \
    }
    if (n_reg > 0) {
        // n_reg is even; the two chunks still in registers after the reloading steps make n_reg + 2 steps, taken in
        // quads (slots 0 .. 3) plus, when n_reg + 2 is not a multiple of four, one final pair on slots 0 / 1
        int left = n_reg;
#pragma unroll 1
        while (left >= 4) {
            BM_STEP_REG(fa, fb, g0, true, 0)
            BM_STEP_REG(fb, fa, g1, true, 1)
            BM_STEP_REG(fa, fb, g0, true, 2)
            BM_STEP_REG(fb, fa, g1, true, 3)
            left -= 4;
        }
        if (left == 2) {           // slots 0, 1 reload; 2, 3 drain
            BM_STEP_REG(fa, fb, g0, true, 0)
            BM_STEP_REG(fb, fa, g1, true, 1)
            BM_STEP_REG(fa, fb, g0, false, 2)
            BM_STEP_REG(fb, fa, g1, false, 3)
        } else {                   // drain on slots 0, 1: the tail below continues at slot 2
            BM_STEP_REG(fa, fb, g0, false, 0)
            BM_STEP_REG(fb, fa, g1, false, 1)
        }
    }
    // leading run of DMA steady steps
    int n_steady = 0;
    if (FAST && STG != STG_REG && !BM_ABL(6) && !BM_ABL(0))
        while (n_steady + PF < nch) {                  // every chunk the step's counted wait assumes in flight is a DMA chunk
            bool all = true;
#pragma unroll
            for (int c_ = 2; c_ <= PF; ++c_) all = all && is_dma(n_steady + c_);
            if (!all) break;
            ++n_steady;
        }
#pragma unroll 1
    for (int q = 0; q < n_steady / 4; ++q) {
        BM_STEP_Q(fa, fb, 0)
        BM_STEP_Q(fb, fa, 1)
        BM_STEP_Q(fa, fb, 2)
        BM_STEP_Q(fb, fa, 3)
    }
    // the remaining steps before the last one.  Compile-time slots again (cc % 4 == 0 here, or == 2 after a REG run
    // that ended on a pair: then the first pass enters the quad in the middle) - except for the 8-wave geometry with
    // two accumulator tiles per wave, whose 256-VGPR budget the four unrolled copies overflow (spills in the hot
    // loop: 32 us instead of 25 for the 784x1024 outer products); it takes these few steps on run-time slots.
    constexpr bool CTS = !(G::NW == 8 && G::MI * G::NJ > 1);
    const int last = nch - 1;
    bool odd = false;                                 // true: the current fragments are in fb
    if (CTS) {
        if ((cc & 3) == 2 && cc < last) {
            BM_STEP_TC(fa, fb, 2)
            if (cc < last) { BM_STEP_TC(fb, fa, 3) } else odd = true;
        }
#pragma unroll 1
        while (cc < last && !odd) {
            BM_STEP_TC(fa, fb, 0)
            if (cc >= last) { odd = true; break; }
            BM_STEP_TC(fb, fa, 1)
            if (cc >= last) break;
            BM_STEP_TC(fa, fb, 2)
            if (cc >= last) { odd = true; break; }
            BM_STEP_TC(fb, fa, 3)
        }
    } else {
#pragma unroll 1
        while (cc < last) {
            BM_STEP_TR(fa, fb)
            if (cc >= last) { odd = true; break; }
            BM_STEP_TR(fb, fa)
        }
    }
    if (odd) fa = fb;            // keep the current fragments in `fa` (once per kernel)
    BM_MSTAMP(3);
    if (last < 1) side.drain();  // (a one-chunk contraction: no step before the last)
    // last chunk: only the k blocks it really holds.  No operand load is in flight any more (the last chunk was
    // waited for two steps ago); the loads of side.drain() are hipcc's to wait for, where the epilogue uses them
    mfma_frags_head<G, ABL>(acc, fa, nq_last);
    if (!BM_ABL(5) && Side::kFinalSync) wg_barrier();
    BM_MSTAMP(4);
#undef BM_STEP_TC
#undef BM_STEP_TR
#undef BM_STEP_T
#undef BM_STEP_Q
#undef BM_DMA_AT
#undef BM_STEP_REG
#undef BM_REG_SCHED
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

// XCD-aware block -> tile map.  Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8,
// MI355X_MICROARCH.md; a speed assumption only - any placement computes the same tiles), every XCD has its own
// 4 MiB L2, and the L2s are refilled from the Infinity Cache after every kernel boundary.  What a launch pulls
// through the fabric is therefore sum over XCDs of (the i-operand panels + the j-operand panels its tiles touch):
//   * the 8 XCDs form an xi x xj grid over the tile matrix; XCD (a, b) owns the rectangle of tiles
//     i in [a tiles_i / xi, (a+1) tiles_i / xi) x j in [b tiles_j / xj, ...) - a 1-D slab (round 2: xi = 8 or xj = 8)
//     makes every L2 read ALL of the other operand (prop-up at 784 x 1024 x 512: 16.3 MB for 4.8 MB of operands);
//   * inside its rectangle an XCD walks j-GROUPS of gj tile columns, all tile rows of a group before the next group,
//     so that the gj j-panels of a group stay L2 resident while the i-panels stream past them (the 3072 x 5000 outer
//     products: 6.3 MB of X / v column panels per slab did not fit the L2 and were re-streamed once per tile row,
//     683 MB per launch for 124 MB of operands).
// xi, xj, gj are chosen on the host by make_tile_map (modelled fill traffic; BM355_DEBUG=xcd_map=xi:gj overrides).
// Rectangle sizes and the number of blocks an XCD receives differ by a few tiles: tiles beyond an XCD's block count
// are handed, in a fixed order, to the XCDs with spare blocks.
struct TileMap {
    int tiles_i, tiles_j, xi, xj, gj;
    int slab;       // 1: the round-2 order (block_to_tile: every XCD a contiguous run of tiles in row-major order, no table):
                    // the choice of the launch tuner unless a grid measures at least 2 % faster - the grids only pay where
                    // an XCD's panels overflow its L2 (3072 x 5000, AIS); at the 784 x 1024 shapes the table walk in the
                    // kernel prologue cost ~0.5 us per launch for nothing (same-box A/B, round 4)
    // per XCD, host-computed and packed so that the device reads them as 12 scalar registers (see tile_of_block):
    //   rect[x]  = i0 | hi << 16 | j0 << 32 | wj << 48      the XCD's rectangle of tiles
    //   spare[x] = spare_before | left << 16                 spare blocks in the XCDs before this one, and the number
    //                                                        of tiles of this rectangle its own blocks do not reach
    unsigned long long rect[8];
    unsigned int spare[8];
};

// force_xi: 0 = the traffic model's choice (or BM355_DEBUG=xcd_map=xi[:gj]); 8 / 4 / 2 / 1 = that grid; -1 = the slab order
// (the launch tuner measures all five per shape)
static inline TileMap make_tile_map(int tiles_i, int tiles_j, double bytes_i, double bytes_j, int force_xi = 0) {
    // bytes_i / bytes_j: operand bytes one tile row / column pulls in (K * tile extent * 4)
    static int env_xi = -1, env_gj = 0;
    if (env_xi < 0) {
        const char *e = bm::dbg("xcd_map");
        env_xi = 0;
        if (e) { env_xi = atoi(e); const char *c = strchr(e, ':'); env_gj = c ? atoi(c + 1) : 0; }
    }
    if (env_xi > 0) force_xi = env_xi;
    const bool slab = force_xi < 0;
    if (slab) force_xi = 0;
    const double l2_budget = 2.5 * 1024 * 1024;          // of 4 MiB: the rest holds the streaming panels and outputs
    TileMap best;
    memset(&best, 0, sizeof(best));
    best.tiles_i = tiles_i; best.tiles_j = tiles_j; best.xi = 8; best.xj = 1; best.gj = tiles_j;
    best.slab = slab ? 1 : 0;
    double best_cost = -1.0;
    for (int xi = 8; xi >= 1; xi >>= 1) {
        const int xj = 8 / xi;
        if (force_xi > 0 && xi != force_xi) continue;
        const double hi = (double)tiles_i / xi, wj = (double)tiles_j / xj;
        // widest j-group whose panels fit the budget next to ~4 streaming i-panels
        int gj = (int)((l2_budget - 4.0 * bytes_i) / (bytes_j > 1.0 ? bytes_j : 1.0));
        if (gj < 1) gj = 1;
        if (gj > (int)(wj + 0.999)) gj = (int)(wj + 0.999);
        if (env_gj > 0) gj = env_gj;
        const double groups = (double)(((int)(wj + 0.999) + gj - 1) / gj);
        const double cost = 8.0 * (hi * bytes_i * groups + wj * bytes_j);
        if (best_cost < 0.0 || cost < best_cost * 0.999) { best_cost = cost; best.xi = xi; best.xj = xj; best.gj = gj; }
    }
    if (best.gj < 1) best.gj = 1;
    const int nb = tiles_i * tiles_j, q = nb / 8, r = nb % 8;
    int spare = 0;
    for (int x = 0; x < 8; ++x) {
        const int a = x % best.xi, c = x / best.xi;
        const int i0 = a * tiles_i / best.xi, hi = (a + 1) * tiles_i / best.xi - i0;
        const int j0 = c * tiles_j / best.xj, wj = (c + 1) * tiles_j / best.xj - j0;
        best.rect[x] = (unsigned long long)(i0 & 0xFFFF) | (unsigned long long)(hi & 0xFFFF) << 16 |
                       (unsigned long long)(j0 & 0xFFFF) << 32 | (unsigned long long)(wj & 0xFFFF) << 48;
        const int cap = q + (x < r ? 1 : 0), sz = hi * wj;
        const int left = sz > cap ? sz - cap : 0;
        best.spare[x] = (unsigned int)(spare & 0xFFFF) | (unsigned int)(left & 0xFFFF) << 16;
        if (cap > sz) spare += cap - sz;
    }
    return best;
}

// a / b for 0 <= a < 2^23, 0 < b < 2^23 without the 32-bit division expansion: float quotient + one-step fix-up
__host__ __device__ __forceinline__ int small_div(int a, int b) {
    int q = (int)((float)a / (float)b);
    q -= (q * b > a);
    q += ((q + 1) * b <= a);
    return q;
}

// e / sp: the words of the block's own XCD (rect[b & 7], spare[b & 7]), already fetched by the caller
__host__ __device__ __forceinline__ void tile_of_block_pre(const TileMap &m, unsigned long long e, unsigned int sp,
                                                           int gj, int b, int nb, int &ti, int &tj) {
    const int q = nb >> 3, r = nb & 7;
    int l = b >> 3;
    int hi = (int)(e >> 16) & 0xFFFF, wj = (int)(e >> 48) & 0xFFFF;
    if (l >= hi * wj) {
        // a spare block of this XCD: its rank among all spare blocks (XCD order, then local order) takes the tile of
        // the same rank among the tiles no block of their own XCD reaches
        int k = l - hi * wj + (int)(sp & 0xFFFF);
#pragma unroll
        for (int xc = 0; xc < 8; ++xc) {
            const int lf = (int)(m.spare[xc] >> 16);
            if (k >= 0 && k < lf) {
                l = q + (xc < r ? 1 : 0) + k;
                e = m.rect[xc];
                k = -1;
            } else if (k >= 0) {
                k -= lf;
            }
        }
        hi = (int)(e >> 16) & 0xFFFF; wj = (int)(e >> 48) & 0xFFFF;
    }
    const int i0 = (int)e & 0xFFFF, j0 = (int)(e >> 32) & 0xFFFF;
    // l-th tile of the rectangle: j-groups of gj columns, inside a group row by row
    const int nfull = small_div(wj, gj), per = hi * gj;
    if (l < nfull * per) {
        const int g = small_div(l, per), rem = l - g * per, row = small_div(rem, gj);
        ti = i0 + row; tj = j0 + g * gj + (rem - row * gj);
    } else {
        const int wl = wj - nfull * gj > 0 ? wj - nfull * gj : 1, rem = l - nfull * per, row = small_div(rem, wl);
        ti = i0 + row; tj = j0 + nfull * gj + (rem - row * wl);
    }
}

// Host evaluation (tests, tools).
static inline void tile_of_block(const TileMap &m, int b, int nb, int &ti, int &tj) {
    tile_of_block_pre(m, m.rect[b & 7], m.spare[b & 7], m.gj, b, nb, ti, tj);
}

// Device: the block's two words are fetched by SCALAR loads with a register offset (blockIdx & 7 is known when the wave
// starts, so they go out with the first batch of kernel arguments; TILE_WORDS right after the argument asm of the
// kernel).  Round 3 kept the per-XCD entries as six arrays of shorts: there is no 16-bit scalar load, so hipcc fetched
// them with DEPENDENT global loads before the first operand load could be issued, ~0.4 us per launch (same-box A/B
// against the round-2 library, round 4; holding all 8 entries in SGPRs instead costs 25 registers and spills).
#define TILE_WORDS(tmap_, blk_)                                                                                      \
    const unsigned long long tile_rect_ = (tmap_).rect[(blk_) & 7];                                                 \
    const unsigned int tile_spare_ = (tmap_).spare[(blk_) & 7];                                                     \
    const int tile_gj_ = (tmap_).gj;                                                                                \
    asm volatile("" :: "s"(tile_rect_), "s"(tile_spare_), "s"(tile_gj_))
#define TILE_OF_BLOCK(tmap_, blk_, nb_, ti_, tj_) \
    tile_of_block_pre((tmap_), tile_rect_, tile_spare_, tile_gj_, (blk_), (nb_), (ti_), (tj_))

// round-2 form (1-D slabs), still used by the kernels without a TileMap argument (free-energy GEMM)
__device__ __forceinline__ void block_to_tile(int tiles_j, int &ti, int &tj, int skip = 0, int trail = 0,
                                              int q_major = 0, int tiles_i = 1) {
    // `skip` leading / `trail` trailing non-tile workgroups in the launch
    const int nb = gridDim.x - skip - trail, b = blockIdx.x - skip;
    const int q = nb / 8, r = nb % 8, xcd = b % 8;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    if (q_major) { tj = t / tiles_i; ti = t % tiles_i; }
    else         { ti = t / tiles_j; tj = t % tiles_j; }
}

}  // namespace bm
