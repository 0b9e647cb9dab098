// bm_bf3.h — "fast-binary" contraction: exact-product bf16 x 3 on the bf16 matrix cores.
//
// SURVEY §7 hard part 4 / north_star ("MFMA bf16/fp32"): whenever one operand of a propagation is a {0,1} bitmap -
// every contraction of a sampling sweep with sample_v_states / sample_h_states on: the Gibbs sweep, the PCD particle
// sweeps, all three GEMMs of an AIS transition (dbm.py:662-694) - the fp32 weight w can be split EXACTLY into three
// bf16 numbers, w = hi + mid + lo (8 + 8 + 8 significand bits), and
//     sum_k w[k] s[k]  =  sum_k hi[k] s[k] + sum_k mid[k] s[k] + sum_k lo[k] s[k]        (s[k] in {0, 1})
// holds term by term: every product of the right-hand side is exact in bf16 x bf16 -> fp32.  Three
// v_mfma_f32_16x16x32_bf16 (one per plane, same fp32 accumulator) replace eight v_mfma_f32_16x16x4_f32 per 32 k, at
// 16x the rate: 5.3x less matrix time.  Only the ORDER of the fp32 additions differs from the canonical chain of
// bm_gemm.h, so results agree with the fp32 path to fp32 round-off (1e-6 relative on the pre-activations), NOT bit
// for bit: this is an opt-in mode (bm_*_set_fast_binary), never the default and never the headline benchmark; its
// parity bar is a tolerance plus a count of the draws that fall on the other side of u < p (tests/test_fast_binary_gpu.py).
//
// Data the mode keeps next to the fp32 state:
//   * weight planes  P3[3][x][ld]  bf16, x-major (k contiguous), one set per contraction direction (W and W^T),
//     rebuilt by split3_kernel when the parameters change;  ld % 64 == 0, zero padded;
//   * state shadows  S16[rows][ld] bf16 (a {0,1} state is exact), written by the producing act_kernel next to the
//     fp32 states (ActArgs::states16), zero padded to ld % 64 == 0 - so K needs no tail handling here.
//
// LDS images: an x-major bf16 tile row of 64 k is 128 bytes = the x-major fp32 tile row of BK = 32 of bm_gemm.h:
// the same 16-byte chunk XOR swizzle (fx<32>) and the same conflict-free ds_read_b128 pattern (a lane's 8 k of one
// MFMA operand are ONE 16-byte chunk), filled by the same LDS-DMA instruction.  The wave tile is the MI = 2 tile of
// the fp32 engine (32 i x 16 j, accumulator layout included), so act_kernel's epilogue is shared unchanged; the P
// image holds the rows of a wave's two MFMA tiles de-interleaved (image row 32 wi + 16 t + m <-> i = 32 wi + 2 m + t).
#pragma once
#include "bm_gemm.h"

namespace bm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Bf3Operand {
    const uint16_t *p;      // element (plane, x, k) at p[plane * plane_stride + x * ld + k]
    long long plane_stride; // elements between planes (P operands; 0 for the state operand)
    int ld;                 // elements per row, multiple of 64, zero padded
    int nx;                 // rows
};
struct Bf3Range {
    Bf3Operand P1, Q1; int K1;      // K in bf16 elements, rounded up to a multiple of 64 (the padding is zero)
    Bf3Operand P2, Q2; int K2;      // K2 == 0: absent
};

template <class G> struct Bf3Geo {
    static_assert(G::MI == 2 && G::NJ == 1, "bf16x3 path: the 32 x 16 wave tile");
    static constexpr int NBUF = 4, PF = 3;
    static constexpr int P_ROWS = 3 * G::TI, Q_ROWS = G::TJ;
    static constexpr int P_FLOATS = P_ROWS * 32, Q_FLOATS = Q_ROWS * 32;          // 128-byte rows
    static constexpr int SLOT_FLOATS = P_FLOATS + Q_FLOATS;
    static constexpr int SMEM_FLOATS = NBUF * SLOT_FLOATS;
    static constexpr int NPP = P_FLOATS * 4 / (1024 * G::NW), NPQ = Q_FLOATS * 4 / (1024 * G::NW);
    static constexpr int NPW = NPP + NPQ;
    static_assert(NPP >= 1 && NPP * 1024 * G::NW == P_FLOATS * 4 && NPQ >= 1 && NPQ * 1024 * G::NW == Q_FLOATS * 4, "whole DMA pieces per wave");
    static_assert(SMEM_FLOATS * 4 <= 160 * 1024 - 1024, "LDS");
};

// operand fragments of one 64-k chunk: a[kb][plane][t], b[kb] - each one 16-byte chunk = 8 bf16
template <class G> struct Bf3Frags { f32x4 a[2][3][2]; f32x4 b[2]; };

template <class G>
__device__ __forceinline__ void bf3_read(Bf3Frags<G> &f, const float *sP, const float *sQ, int wi, int wj, int lane) {
    const int g = lane >> 4, l15 = lane & 15;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = p * G::TI + wi * 32 + 16 * t + l15;
                f.a[kb][p][t] = *reinterpret_cast<const f32x4 *>(sP + img_off<XM, 0, 32, 0>(row, 4 * kb + g));
            }
        f.b[kb] = *reinterpret_cast<const f32x4 *>(sQ + img_off<XM, 0, 32, 0>(wj * 16 + l15, 4 * kb + g));
    }
}

template <class G>
__device__ __forceinline__ void bf3_mfma(f32x4 (&acc)[2][1], const Bf3Frags<G> &f) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.a[kb][p][t]),
                                                                    __builtin_bit_cast(bf16x8, f.b[kb]), acc[t][0], 0, 0, 0);
}

// acc[t] += sum_k P[i][k] Q[j][k] over segment 1 then segment 2; same ring discipline as bm_gemm.h mainloop (chunk c
// in slot c % 4, DMA three chunks ahead, counted vmcnt + raw barrier, fragments one chunk ahead in registers), split
// into phases so that a persistent workgroup can request the first chunks of its NEXT tile before the epilogue of the
// current one: setup (once: the plan of the weight planes) / set_tile (the plan of the state rows) / prefetch / run.
template <class G, bool SEG2> struct Bf3Pipe {
    using B = Bf3Geo<G>;
    static constexpr int NBUF = B::NBUF, PF = B::PF, NPW = B::NPW;
    uint32_t p1[B::NPP], p2[B::NPP], q1[B::NPQ], q2[B::NPQ];
    const char *P1, *P2, *Q1, *Q2;
    float *smem;
    unsigned ldsPw, ldsQw;
    int nch1, nch, w, wi, wj, lane;

    __device__ __forceinline__ void setup(const Bf3Range &kr, int i0, float *smem_) {
        smem = smem_;
        const int tid = threadIdx.x;
        lane = tid & 63;
        w = __builtin_amdgcn_readfirstlane(tid >> 6);
        wi = w % G::WI; wj = w / G::WI;
        nch1 = kr.K1 / 64; nch = nch1 + (SEG2 ? kr.K2 / 64 : 0);
        P1 = (const char *)kr.P1.p; Q1 = (const char *)kr.Q1.p;
        P2 = SEG2 ? (const char *)kr.P2.p : P1; Q2 = SEG2 ? (const char *)kr.Q2.p : Q1;
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
        ldsPw = lds0 + (unsigned)w * 1024u; ldsQw = lds0 + (unsigned)(B::P_FLOATS * 4) + (unsigned)w * 1024u;
        plan_p(p1, kr.P1, i0);
        if (SEG2) plan_p(p2, kr.P2, i0);
    }
    __device__ __forceinline__ void plan_p(uint32_t (&o)[B::NPP], const Bf3Operand &P, int i0) {
#pragma unroll
        for (int n = 0; n < B::NPP; ++n) {
            const int row = (w + n * G::NW) * 8 + (lane >> 3), slot = lane & 7;        // image row / 16-byte slot
            const int plane = row / G::TI, rr = row % G::TI;
            const int i = 32 * (rr >> 5) + 2 * (rr & 15) + ((rr >> 4) & 1);            // de-interleaved wave tiles (header)
            const int c4 = slot ^ fx<32>(row);
            o[n] = (uint32_t)(((long long)plane * P.plane_stride + (long long)min(i0 + i, P.nx - 1) * P.ld + c4 * 8) * 2);
        }
    }
    __device__ __forceinline__ void plan_q(uint32_t (&o)[B::NPQ], const Bf3Operand &Q, int j0) {
#pragma unroll
        for (int n = 0; n < B::NPQ; ++n) {
            const int row = (w + n * G::NW) * 8 + (lane >> 3), slot = lane & 7;
            const int c4 = slot ^ fx<32>(row);
            o[n] = (uint32_t)(((long long)min(j0 + row, Q.nx - 1) * Q.ld + c4 * 8) * 2);
        }
    }
    __device__ __forceinline__ void set_tile(const Bf3Range &kr, int j0) {
        plan_q(q1, kr.Q1, j0);
        if (SEG2) plan_q(q2, kr.Q2, j0);
    }
    __device__ __forceinline__ void dma(int c, int slot) {                  // chunk c (wave-uniform) -> ring slot
        const bool s2 = SEG2 && c >= nch1;
        const int kc = s2 ? c - nch1 : c;
        const char *pb = (s2 ? P2 : P1) + (size_t)kc * 128, *qb = (s2 ? Q2 : Q1) + (size_t)kc * 128;
        const unsigned so = (unsigned)(slot * B::SLOT_FLOATS * 4);
        if (s2) {
#pragma unroll
            for (int n = 0; n < B::NPP; ++n) dma16s(pb, p2[n], ldsPw + so + (unsigned)(n * G::NW * 1024));
#pragma unroll
            for (int n = 0; n < B::NPQ; ++n) dma16s(qb, q2[n], ldsQw + so + (unsigned)(n * G::NW * 1024));
        } else {
#pragma unroll
            for (int n = 0; n < B::NPP; ++n) dma16s(pb, p1[n], ldsPw + so + (unsigned)(n * G::NW * 1024));
#pragma unroll
            for (int n = 0; n < B::NPQ; ++n) dma16s(qb, q1[n], ldsQw + so + (unsigned)(n * G::NW * 1024));
        }
    }
    // chunks 0 .. PF-1 of the current tile -> slots 0 .. PF-1.  Every slot is free: the last fragment read of the
    // previous tile was followed by a workgroup barrier inside run().
    __device__ __forceinline__ void prefetch() {
#pragma unroll
        for (int c = 0; c < PF; ++c) if (c < nch) dma(c, c);
    }
    __device__ __forceinline__ void run(f32x4 (&acc)[2][1]) {
        // everything this wave has in flight - the prefetched chunks, and the stores of the previous tile's epilogue,
        // which share the counter - has landed
        BM_WAIT_VM(0);
        wg_barrier();
        Bf3Frags<G> fa, fb;
        bf3_read<G>(fa, smem, smem + B::P_FLOATS, wi, wj, lane);
        if (nch == 1) wg_barrier();      // (no step follows: the next tile's prefetch must not overtake this read)
        int cc = 0;
        // step: DMA of chunk cc+3, fragments of chunk cc+1, MFMAs of chunk cc; S = cc % 4 is a compile-time constant.
        // In the first two steps chunk cc+2 has already landed (run() waited for the whole prefetch); from then on
        // only the DMA issued in the same step may stay in flight.
#define BM_BF3_STEP(FC, FN, S)                                                                           \
    {                                                                                                    \
        if (cc + PF < nch) dma(cc + PF, ((S) + PF) % NBUF);                                              \
        bf3_read<G>(FN, smem + (((S) + 1) % NBUF) * B::SLOT_FLOATS, smem + (((S) + 1) % NBUF) * B::SLOT_FLOATS + B::P_FLOATS, wi, wj, lane); \
        bf3_mfma<G>(acc, FC);                                                                            \
        if (cc + PF < nch) BM_WAIT_VM(NPW); else BM_WAIT_VM(0);      /* chunk cc+2 has landed */           \
        wg_barrier();                                                                                    \
        ++cc;                                                                                            \
    }
        const int last = nch - 1;
#pragma unroll 1
        while (cc + 4 <= last) {
            BM_BF3_STEP(fa, fb, 0)
            BM_BF3_STEP(fb, fa, 1)
            BM_BF3_STEP(fa, fb, 2)
            BM_BF3_STEP(fb, fa, 3)
        }
        bool odd = false;
        if (cc < last) { BM_BF3_STEP(fa, fb, 0) odd = true; }
        if (cc < last) { BM_BF3_STEP(fb, fa, 1) odd = false; }
        if (cc < last) { BM_BF3_STEP(fa, fb, 2) odd = true; }
#undef BM_BF3_STEP
        if (odd) bf3_mfma<G>(acc, fb); else bf3_mfma<G>(acc, fa);
    }
};

// fp32 matrix -> three bf16 planes (exact: w = hi + mid + lo), x-major with k contiguous.
//   transpose == 0: out[plane][r][c] = split(W[r][c])  (rows of W are the x of the operand, columns its k)
//   transpose == 1: out[plane][c][r] = split(W[r][c])
// The padding beyond the 8-k group that holds the last k (up to ld) is left as the caller zero-initialised it; inside
// that group it is written as zero.
// One thread splits 8 consecutive k of one x and stores three 16-byte chunks (the planes' rows are 128-byte aligned);
// for the transposed build adjacent threads take adjacent columns of W, so the 8 strided reads stay coalesced.
__device__ __forceinline__ void split3(float w, uint32_t &h, uint32_t &m, uint32_t &l) {
    const uint32_t hb = __float_as_uint(w) & 0xffff0000u;
    const float r1 = w - __uint_as_float(hb);
    const uint32_t mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    h = hb >> 16; m = mb >> 16; l = __float_as_uint(r2) >> 16;
}
template <bool VEC>
__device__ __forceinline__ void split3_fetch(const float *W, int ldw, int nx, int nk, int kg, int transpose, size_t e, float (&w)[8]) {
    // transposed: x fastest across threads (coalesced reads of W rows); plain: k-group fastest (coalesced both ways)
    const int x = transpose ? (int)(e % (size_t)nx) : (int)(e / (size_t)kg);
    const int k0 = 8 * (transpose ? (int)(e / (size_t)nx) : (int)(e % (size_t)kg));
    if (VEC) {                                   // plain layout, whole 16-byte aligned groups only
        const float4 a = *reinterpret_cast<const float4 *>(W + (size_t)x * ldw + k0);
        const float4 b = *reinterpret_cast<const float4 *>(W + (size_t)x * ldw + k0 + 4);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    } else {                                     // clamped addresses, no branch around a load
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + q, kc = k < nk ? k : nk - 1;
            const float v = transpose ? W[(size_t)kc * ldw + x] : W[(size_t)x * ldw + kc];
            w[q] = k < nk ? v : 0.f;
        }
    }
}
__device__ __forceinline__ void split3_emit(const float (&w)[8], uint16_t *out, long long plane_stride, int ld, int nx, int kg, int transpose, size_t e) {
    const int x = transpose ? (int)(e % (size_t)nx) : (int)(e / (size_t)kg);
    const int k0 = 8 * (transpose ? (int)(e / (size_t)nx) : (int)(e % (size_t)kg));
    uint32_t h[8], m[8], l[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) split3(w[q], h[q], m[q], l[q]);
    uint16_t *o = out + (size_t)x * ld + k0;                                 // ld % 64 == 0, k0 % 8 == 0: 16-byte aligned
    *reinterpret_cast<uint4 *>(o) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    *reinterpret_cast<uint4 *>(o + plane_stride) = make_uint4(m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16));
    *reinterpret_cast<uint4 *>(o + 2 * plane_stride) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}
template <bool VEC>
__device__ __forceinline__ void split3_body(const float *W, int ldw, int nx, int nk, uint16_t *out, long long plane_stride, int ld, int transpose) {
    const int kg = (nk + 7) >> 3;
    const size_t n = (size_t)nx * kg, stride = (size_t)gridDim.x * blockDim.x;
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; e + stride < n; e += 2 * stride) {    // two groups per trip: four (sixteen) loads in flight per thread
        float w0[8], w1[8];
        split3_fetch<VEC>(W, ldw, nx, nk, kg, transpose, e, w0);
        split3_fetch<VEC>(W, ldw, nx, nk, kg, transpose, e + stride, w1);
        split3_emit(w0, out, plane_stride, ld, nx, kg, transpose, e);
        split3_emit(w1, out, plane_stride, ld, nx, kg, transpose, e + stride);
    }
    if (e < n) {
        float w0[8];
        split3_fetch<VEC>(W, ldw, nx, nk, kg, transpose, e, w0);
        split3_emit(w0, out, plane_stride, ld, nx, kg, transpose, e);
    }
}
__global__ __launch_bounds__(256) void split3_kernel(const float *W, int ldw, int rows, int cols, uint16_t *out, long long plane_stride, int ld, int transpose) {
    const int nx = transpose ? cols : rows, nk = transpose ? rows : cols;      // operand rows / contraction length
    const bool vec = !transpose && (ldw & 3) == 0 && (((uintptr_t)W & 15u) == 0) && (nk & 7) == 0;
    if (vec) split3_body<true>(W, ldw, nx, nk, out, plane_stride, ld, transpose);
    else     split3_body<false>(W, ldw, nx, nk, out, plane_stride, ld, transpose);
}

// fp32 {0,1} states -> bf16 shadow (for states that were not produced by an act_kernel: AIS x_0, user input)
__global__ void shadow16_kernel(const float *X, int ldx, int rows, int cols, uint16_t *out, int ld) {
    const size_t n = (size_t)rows * cols;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / (size_t)cols, c = e % (size_t)cols;
        out[r * ld + c] = (uint16_t)(__float_as_uint(X[r * ldx + c]) >> 16);
    }
}

// the same for caller-provided states: anything that is not exactly 0.0f or 1.0f raises *bad
__global__ void shadow16_check_kernel(const float *X, int ldx, int rows, int cols, uint16_t *out, int ld, int *bad) {
    const size_t n = (size_t)rows * cols;
    bool any = false;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / (size_t)cols, c = e % (size_t)cols;
        const float x = X[r * ldx + c];
        any = any || !(x == 0.0f || x == 1.0f);
        out[r * ld + c] = (uint16_t)(__float_as_uint(x) >> 16);
    }
    if (any) *bad = 1;
}

}  // namespace bm
