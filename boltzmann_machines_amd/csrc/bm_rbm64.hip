// bm_rbm64.hip — the RBM hot path in float64 (bm_rbm64_* entry points of include/bm355.h).
//
// The reference's dtype is a constructor argument (base/mixin.py:15) and its own tests train a
// float64 BernoulliRBM (rbm/tests/test_rbm.py:53-56,70-73).  This is the path for that dtype: the same graph
// (base_rbm.py:415-531), every operation in IEEE double, every dot product the sequential ascending-k fma
// chain of the float64 functions of oracle/bm_oracle.c (bit-identical).
// The contractions run on the FP64 matrix cores (v_mfma_f64_16x16x4_f64: an exact fma chain over its four k,
// continued through the accumulator): one 16 x 16 output tile per wave, operands straight from L1 / L2 (a W
// panel is read coalesced in 128-byte rows, the 16 input rows of a tile are shared by the four waves of a
// workgroup).  Round 1 used one thread per output and vector FMAs (0.62 ms per 784x1024x512 CD-1 update).
// Bernoulli hidden units; Bernoulli or Gaussian visible units.
#include "bm_common.h"
#include "bm_rng.h"

namespace bm64 {
using bm::PhiloxKey;

// TF Uint64ToDouble: 52 mantissa bits from a word pair, [1,2) - 1
__device__ __forceinline__ double u64_to_uniform(uint32_t x0, uint32_t x1) {
    const unsigned long long u = (1023ull << 52) | (((unsigned long long)x0 & 0xfffffull) << 32) | (unsigned long long)x1;
    return __longlong_as_double((long long)u) - 1.0;
}
// element idx of a float64 stream: 2 per Philox block
__device__ __forceinline__ double uniform_at(const PhiloxKey &key, unsigned long long idx) {
    uint32_t w[4];
    bm::philox_block(key, idx >> 1, w);
    return (idx & 1) ? u64_to_uniform(w[2], w[3]) : u64_to_uniform(w[0], w[1]);
}
__device__ __forceinline__ double normal_at(const PhiloxKey &key, unsigned long long idx) {
    uint32_t w[4];
    bm::philox_block(key, idx >> 1, w);
    double u1 = u64_to_uniform(w[0], w[1]);
    if (u1 < 1.0e-20) u1 = 1.0e-20;
    const double v1 = 6.283185307179586476925286766559 * u64_to_uniform(w[2], w[3]);
    const double r = sqrt(-2.0 * log(u1));
    return (idx & 1) ? cos(v1) * r : sin(v1) * r;
}

// exp(-a), a in [0, 700]: Cody-Waite + degree-13 Taylor in Horner form, fma only (oracle: exp_neg_d)
__device__ __forceinline__ double exp_neg(double a) {
    const double t = a * -1.4426950408889634074;
    const double n = rint(t);
    double r = fma(n, -6.93147180369123816490e-01, -a);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const long long ni = (long long)n;
    return __longlong_as_double(__double_as_longlong(p) + (ni << 52));
}
__device__ __forceinline__ double sigmoid(double x) {
    double a = fabs(x);
    if (a > 700.0) a = 700.0;
    const double e = exp_neg(a);
    const double d = 1.0 + e;
    return (x >= 0.0) ? (1.0 / d) : (e / d);
}
__device__ __forceinline__ double softplus(double x) { return fmax(x, 0.0) + log1p(exp(-fabs(x))); }

// z[j][i] = sum_k P[k][i] * Q[j][k] (k ascending fma chain), then the layer activation / draw.
// One wave = 64 consecutive i of one row j: P is read coalesced, Q[j][k] is wave-uniform.
struct ActArgs {
    const double *P; int ldp;        // k-major [K][I]
    const double *Q; int ldq;        // rows [J][K]
    int K, I, J;
    const double *bias, *sigma;
    double mult;
    int kind, sample;
    double *means, *states;          // dense [J][I], may be null
    PhiloxKey key; long long row0;
};
typedef double d4 __attribute__((ext_vector_type(4)));

// ---- FP64-MFMA tile engine (float64 twin of csrc/bm_gemm.h, much simpler: the float64 path is a compatibility
// path, it only has to stay within a small factor of the float32 one).
// Workgroup = 4 waves = 64 (i) x 32 (j) outputs; wave (wi, wj) = 32 x 16 = two 16 x 16 tiles that share the B
// fragment.  K streams in chunks of 32 through a double-buffered LDS tile pair filled through registers.
//   v_mfma_f64_16x16x4_f64:  A (lane l: m = l & 15, k = l >> 4), B (lane l: n = l & 15, k = l >> 4),
//   D (lane l, register r) = (m = 4 r + (l >> 4), n = l & 15)          [tools/mf64_probe.hip]
// A <- P[k][i] (k-major), B <- Q: x-major rows [j][k] (propagations) or k-major [k][j] (outer products).
// The four k of an instruction and the instructions of a dot product run in ascending k: the sequential
// chain of the float64 oracle.
constexpr int T64_TI = 64, T64_TJ = 32, T64_BK = 32;
constexpr int T64_PLD = T64_TI + 16;        // k-major P rows: +128 B, rows k / k+1 hit different bank halves
constexpr int T64_QLD_XM = T64_BK + 2;      // x-major Q rows: +16 B per row
constexpr int T64_QLD_KM = T64_TJ + 16;     // k-major Q rows (32 j + pad)
constexpr int T64_PBUF = T64_BK * T64_PLD;                                   // doubles
constexpr int T64_QBUF = (T64_TJ * T64_QLD_XM > T64_BK * T64_QLD_KM) ? T64_TJ * T64_QLD_XM : T64_BK * T64_QLD_KM;

// [rows][cols] window of a row-major matrix -> registers (pairs of doubles), zero outside [nrows][ncols]
template <int ROWS, int COLS>
__device__ __forceinline__ void t64_fetch(double2 (&r)[ROWS * COLS / 512], const double *p, int ld, int row0, int col0,
                                          int nrows, int ncols, int tid) {
    constexpr int NV = ROWS * COLS / 512, CP = COLS / 2;
    // branch-free: clamped addresses, unconditional loads (all of a chunk's loads are in flight together),
    // out-of-range elements selected to zero afterwards
    const bool vec = ((ld & 1) == 0) && (((uintptr_t)p & 15u) == 0) && ncols >= 2;      // wave-uniform (col is even)
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int f = tid + n * 256;
        const int row = row0 + f / CP, col = col0 + 2 * (f % CP);
        const int rc = min(row, nrows - 1);
        double2 v;
        if (vec) {
            v = *reinterpret_cast<const double2 *>(p + (size_t)rc * ld + min(col, (ncols - 2) & ~1));
        } else {
            v.x = p[(size_t)rc * ld + min(col, ncols - 1)];
            v.y = p[(size_t)rc * ld + min(col + 1, ncols - 1)];
        }
        const bool rok = row < nrows;
        r[n] = make_double2((rok && col < ncols) ? v.x : 0.0, (rok && col + 1 < ncols) ? v.y : 0.0);
    }
}
template <int ROWS, int COLS, int LD>
__device__ __forceinline__ void t64_stash(const double2 (&r)[ROWS * COLS / 512], double *s, int tid, double sgn) {
    constexpr int NV = ROWS * COLS / 512, CP = COLS / 2;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int f = tid + n * 256;
        double *d = s + (f / CP) * LD + 2 * (f % CP);
        d[0] = sgn * r[n].x; d[1] = sgn * r[n].y;
    }
}

struct T64Seg { const double *P; int ldp; const double *Q; int ldq; int K; double qsgn; };

// ---- interior tiles of 16-byte-aligned operands: the full K chunks of a segment stream through a pipeline without
// address arithmetic in the loop (the lesson of the float32 engine: vector ALU work between the MFMAs of a
// dependent chain is paid for in matrix-pipe time).  Per-thread byte offsets are fixed, the chunk base is scalar,
// two register sets keep two chunks in flight, the loop is unrolled by two so that the LDS buffer is a constant.
struct T64Fast {
    uint32_t op[4], oq[2];       // byte offsets of this thread's double2 loads relative to the chunk bases
    int sp[4], sq[2];            // LDS double offsets of the same pieces inside a buffer
};
template <bool QKM>
__device__ __forceinline__ void t64_fast_plan(T64Fast &f, int ldp, int ldq, int tid) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int row = tid / 32 + 8 * n, col = 2 * (tid % 32);
        f.op[n] = (uint32_t)(row * ldp + col) * 8u;
        f.sp[n] = row * T64_PLD + col;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int row = tid / 16 + 16 * n, col = 2 * (tid % 16);
        f.oq[n] = (uint32_t)(row * ldq + col) * 8u;
        f.sq[n] = row * (QKM ? T64_QLD_KM : T64_QLD_XM) + col;
    }
}
typedef double dv2 __attribute__((ext_vector_type(2)));
struct T64Regs { dv2 p0, p1, p2, p3, q0, q1; };     // named members, native vectors: stay in registers
__device__ __forceinline__ void t64_fast_fetch(T64Regs &r, const T64Fast &f, const char *pb, const char *qb) {
    r.p0 = *reinterpret_cast<const dv2 *>(pb + f.op[0]); r.p1 = *reinterpret_cast<const dv2 *>(pb + f.op[1]);
    r.p2 = *reinterpret_cast<const dv2 *>(pb + f.op[2]); r.p3 = *reinterpret_cast<const dv2 *>(pb + f.op[3]);
    r.q0 = *reinterpret_cast<const dv2 *>(qb + f.oq[0]); r.q1 = *reinterpret_cast<const dv2 *>(qb + f.oq[1]);
}
__device__ __forceinline__ void t64_fast_stash(const T64Regs &r, const T64Fast &f, double *sP, double *sQ, bool neg) {
    *reinterpret_cast<dv2 *>(sP + f.sp[0]) = r.p0; *reinterpret_cast<dv2 *>(sP + f.sp[1]) = r.p1;
    *reinterpret_cast<dv2 *>(sP + f.sp[2]) = r.p2; *reinterpret_cast<dv2 *>(sP + f.sp[3]) = r.p3;
    // block-uniform: the negated operand of the second segment
    *reinterpret_cast<dv2 *>(sQ + f.sq[0]) = neg ? -r.q0 : r.q0;
    *reinterpret_cast<dv2 *>(sQ + f.sq[1]) = neg ? -r.q1 : r.q1;
}
template <bool QKM>
__device__ __forceinline__ void t64_chunk_mfma(d4 (&acc)[2], const double *sP, const double *sQ, int wi, int wj, int l15, int g) {
    const double *pP = sP + wi * 32 + l15;
    const double *pQ = QKM ? sQ + wj * 16 + l15 : sQ + (wj * 16 + l15) * T64_QLD_XM;
    double a0[T64_BK / 4], a1[T64_BK / 4], bq[T64_BK / 4];
#pragma unroll
    for (int s4 = 0; s4 < T64_BK / 4; ++s4) {
        const int k = 4 * s4 + g;
        a0[s4] = pP[k * T64_PLD]; a1[s4] = pP[k * T64_PLD + 16];
        bq[s4] = QKM ? pQ[k * T64_QLD_KM] : pQ[k];
    }
#pragma unroll
    for (int s4 = 0; s4 < T64_BK / 4; ++s4) {
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s4], bq[s4], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s4], bq[s4], acc[1], 0, 0, 0);
    }
}
// the first `nfull` (>= 1) chunks of one segment; on return the LDS buffers are free again (trailing barrier)
template <bool QKM>
__device__ __forceinline__ void t64_fast_segment(d4 (&acc)[2], const T64Seg &sg, int nfull, int i0, int j0, double *smem, int tid) {
    const int lane = tid & 63, w = tid >> 6;
    const int wi = w & 1, wj = w >> 1, l15 = lane & 15, g = lane >> 4;
    double *sP0 = smem, *sP1 = smem + T64_PBUF, *sQ0 = smem + 2 * T64_PBUF, *sQ1 = smem + 2 * T64_PBUF + T64_QBUF;
    T64Fast f;
    t64_fast_plan<QKM>(f, sg.ldp, sg.ldq, tid);
    const bool neg = sg.qsgn < 0.0;
    const char *pb = (const char *)(sg.P + i0);
    const char *qb = QKM ? (const char *)(sg.Q + j0) : (const char *)(sg.Q + (size_t)j0 * sg.ldq);
    const size_t stp = (size_t)T64_BK * sg.ldp * 8, stq = QKM ? (size_t)T64_BK * sg.ldq * 8 : (size_t)T64_BK * 8;
    T64Regs ra, rb;
    t64_fast_fetch(ra, f, pb, qb); pb += stp; qb += stq;                       // chunk 0
    if (nfull > 1) { t64_fast_fetch(rb, f, pb, qb); pb += stp; qb += stq; }    // chunk 1
    __syncthreads();                                    // the previous user of the buffers is done
    t64_fast_stash(ra, f, sP0, sQ0, neg);
    __syncthreads();
    int c = 0;
    // pairs: chunk c in buffer 0 (loaded through ra), chunk c+1 through rb into buffer 1
    for (; c + 1 < nfull; c += 2) {
        if (c + 2 < nfull) { t64_fast_fetch(ra, f, pb, qb); pb += stp; qb += stq; }     // chunk c+2
        t64_chunk_mfma<QKM>(acc, sP0, sQ0, wi, wj, l15, g);                               // chunk c
        t64_fast_stash(rb, f, sP1, sQ1, neg);                                             // chunk c+1
        __syncthreads();
        if (c + 3 < nfull) { t64_fast_fetch(rb, f, pb, qb); pb += stp; qb += stq; }     // chunk c+3
        t64_chunk_mfma<QKM>(acc, sP1, sQ1, wi, wj, l15, g);                               // chunk c+1
        if (c + 2 < nfull) t64_fast_stash(ra, f, sP0, sQ0, neg);                          // chunk c+2
        __syncthreads();
    }
    if (c < nfull) {                                    // odd count: the last chunk sits in buffer 0
        t64_chunk_mfma<QKM>(acc, sP0, sQ0, wi, wj, l15, g);
        __syncthreads();
    }
}

// acc[t] += sum_k P[k][i] * Q(j, k) for the wave's two tiles (t = 0, 1: i sub-tile), over `nseg` segments
template <bool QKM>
__device__ __forceinline__ void t64_mainloop(d4 (&acc)[2], const T64Seg *seg, int nseg, int I, int J, int i0, int j0,
                                             double *smem) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w & 1, wj = w >> 1, l15 = lane & 15, g = lane >> 4;
    double *sP = smem, *sQ = smem + 2 * T64_PBUF;
    for (int sgi = 0; sgi < nseg; ++sgi) {
        const T64Seg sg = seg[sgi];
        const int nch = (sg.K + T64_BK - 1) / T64_BK;
        // interior tile, 16-byte-legal operands (block-uniform): the full chunks take the pipelined path, the
        // generic loop below starts at the K tail (same k order, same zero fill: bit-identical)
        int c0 = 0;
        const bool fast = i0 + T64_TI <= I && j0 + T64_TJ <= J && ((sg.ldp | sg.ldq) & 1) == 0 &&
                          ((((uintptr_t)sg.P | (uintptr_t)sg.Q) & 15u) == 0) && ((i0 | j0) & 1) == 0;
        if (fast && sg.K >= T64_BK) {
            c0 = sg.K / T64_BK;
            t64_fast_segment<QKM>(acc, sg, c0, i0, j0, smem, tid);
            if (c0 == nch) continue;
        }
        double2 rp[T64_BK * T64_TI / 512], rq[T64_BK * T64_TJ / 512];
        t64_fetch<T64_BK, T64_TI>(rp, sg.P, sg.ldp, c0 * T64_BK, i0, sg.K, I, tid);
        if (QKM) t64_fetch<T64_BK, T64_TJ>(rq, sg.Q, sg.ldq, c0 * T64_BK, j0, sg.K, J, tid);
        else     t64_fetch<T64_TJ, T64_BK>(rq, sg.Q, sg.ldq, j0, c0 * T64_BK, J, sg.K, tid);
        __syncthreads();                               // the previous segment's last chunk has been consumed
        t64_stash<T64_BK, T64_TI, T64_PLD>(rp, sP, tid, 1.0);
        if (QKM) t64_stash<T64_BK, T64_TJ, T64_QLD_KM>(rq, sQ, tid, sg.qsgn);
        else     t64_stash<T64_TJ, T64_BK, T64_QLD_XM>(rq, sQ, tid, sg.qsgn);
        __syncthreads();
        for (int c = c0; c < nch; ++c) {
            const int cur = (c - c0) & 1;
            if (c + 1 < nch) {                          // next chunk in flight under the MFMAs
                t64_fetch<T64_BK, T64_TI>(rp, sg.P, sg.ldp, (c + 1) * T64_BK, i0, sg.K, I, tid);
                if (QKM) t64_fetch<T64_BK, T64_TJ>(rq, sg.Q, sg.ldq, (c + 1) * T64_BK, j0, sg.K, J, tid);
                else     t64_fetch<T64_TJ, T64_BK>(rq, sg.Q, sg.ldq, j0, (c + 1) * T64_BK, J, sg.K, tid);
            }
            const double *pP = sP + cur * T64_PBUF + wi * 32 + l15;
            const double *pQ = QKM ? sQ + cur * T64_QBUF + wj * 16 + l15
                                   : sQ + cur * T64_QBUF + (wj * 16 + l15) * T64_QLD_XM;
            const int ksteps = ((sg.K - c * T64_BK < T64_BK) ? (sg.K - c * T64_BK + 3) / 4 : T64_BK / 4);
            // all fragments of the chunk first (one LDS round trip per chunk, not one per k-step), then its MFMAs
            double a0[T64_BK / 4], a1[T64_BK / 4], bq[T64_BK / 4];
#pragma unroll
            for (int s4 = 0; s4 < T64_BK / 4; ++s4) {
                const int k = 4 * s4 + g;
                a0[s4] = pP[k * T64_PLD]; a1[s4] = pP[k * T64_PLD + 16];
                bq[s4] = QKM ? pQ[k * T64_QLD_KM] : pQ[k];
            }
            if (ksteps == T64_BK / 4) {                 // full chunk: one straight run of MFMAs (wave-uniform test)
#pragma unroll
                for (int s4 = 0; s4 < T64_BK / 4; ++s4) {
                    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s4], bq[s4], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s4], bq[s4], acc[1], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int s4 = 0; s4 < T64_BK / 4; ++s4) {
                    if (s4 < ksteps) {                  // K tail (zero-filled in LDS past K)
                        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s4], bq[s4], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s4], bq[s4], acc[1], 0, 0, 0);
                    }
                }
            }
            if (c + 1 < nch) {
                t64_stash<T64_BK, T64_TI, T64_PLD>(rp, sP + (cur ^ 1) * T64_PBUF, tid, 1.0);
                if (QKM) t64_stash<T64_BK, T64_TJ, T64_QLD_KM>(rq, sQ + (cur ^ 1) * T64_QBUF, tid, sg.qsgn);
                else     t64_stash<T64_TJ, T64_BK, T64_QLD_XM>(rq, sQ + (cur ^ 1) * T64_QBUF, tid, sg.qsgn);
                __syncthreads();
            }
        }
    }
}

__global__ __launch_bounds__(256, 1) void act_kernel(ActArgs a) {
    __shared__ __attribute__((aligned(16))) double smem[2 * (T64_PBUF + T64_QBUF)];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wi = w & 1, wj = w >> 1, l15 = lane & 15, g = lane >> 4;
    const int i0 = blockIdx.x * T64_TI, j0 = blockIdx.y * T64_TJ;
    d4 acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    T64Seg sg = {a.P, a.ldp, a.Q, a.ldq, a.K, 1.0};
    t64_mainloop<false>(acc, &sg, 1, a.I, a.J, i0, j0, smem);
    const int j = j0 + wj * 16 + l15;
    if (j >= a.J) return;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + wi * 32 + 16 * t + 4 * r + g;
            if (i >= a.I) continue;
            const double b = a.mult * a.bias[i];
            const double x = a.mult * acc[t][r];
            const double m = (a.kind == BM_UNIT_BERNOULLI) ? sigmoid(x + b) : (a.kind == 3 ? x + b : (x * a.sigma[i] + b));
            double s = m;
            if (a.sample) {
                const unsigned long long idx = (unsigned long long)(a.row0 + j) * (unsigned long long)a.I + (unsigned long long)i;
                if (a.kind == BM_UNIT_BERNOULLI) s = (uniform_at(a.key, idx) < m) ? 1.0 : 0.0;
                else s = normal_at(a.key, idx) * a.sigma[i] + m;
            }
            if (a.means) a.means[(size_t)j * a.I + i] = m;
            if (a.states) a.states[(size_t)j * a.I + i] = s;
        }
}

// column sums (sequential over rows) + bias / q_means update (base_rbm.py:450-474)
struct BiasArgs {
    const double *X, *vs, *h0m, *hm; int ldx, B, V, H;
    double *vb, *dvb, *hb, *dhb, *q, *pen;
    double N, lr, mom, damping, cost, target;
};
__device__ __forceinline__ void bias_wave(const BiasArgs &a, int blk, int lane) {
    // One wave per 16 columns.  The column sums are the sequential row-ascending chains of the oracle, computed on
    // the FP64 matrix core against a matrix of ones: D[c][*] = sum_b A[c][b] * 1, i.e. acc = fma(a_b, 1, acc) =
    // acc + a_b rounded once, rows in ascending order (the four k of an instruction run in order, as in the GEMMs).
    // A(c, b) = X[b][c] - v[b][c] for the visible columns (the difference is rounded first, as in the oracle's
    // s + (x - v)); for the hidden columns two chains: sum (h0 - h) and sum h.
    const int l15 = lane & 15, g = lane >> 4;
    const int c0 = blk * 16;
    const bool vis = c0 < a.V;                                   // blocks never straddle: the hidden blocks start at
    const int nvb = (a.V + 15) / 16;                             // block nvb (wave-uniform)
    const int hc0 = (blk - nvb) * 16;
    const int c = vis ? c0 + l15 : hc0 + l15;                    // this lane's A row = column
    const int ncol = vis ? a.V : a.H;
    const bool ok = c < ncol;
    const double *pa = vis ? a.X : a.h0m, *pb = vis ? a.vs : a.hm;
    const int lda = vis ? a.ldx : a.H, ldb = vis ? a.V : a.H;
    d4 s1 = {0.0, 0.0, 0.0, 0.0}, s2 = {0.0, 0.0, 0.0, 0.0};
    constexpr int UB = 8;                                        // MFMA steps (4 rows each) whose loads go out together
    const int cc = ok ? c : 0;
    for (int b0 = 0; b0 < a.B; b0 += 4 * UB) {
        double x[UB], y[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int b = b0 + 4 * u + g;
            const int bc = b < a.B ? b : a.B - 1;
            const double xa = pa[(size_t)bc * lda + cc], xb = pb[(size_t)bc * ldb + cc];
            const bool live = ok && b < a.B;
            x[u] = live ? xa - xb : 0.0;                         // padding rows / columns add fma(0, 1, acc) = acc
            y[u] = live ? xb : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (b0 + 4 * u < a.B) {                              // wave-uniform
                s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u], 1.0, s1, 0, 0, 0);
                if (!vis) s2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y[u], 1.0, s2, 0, 0, 0);
            }
        }
    }
    if (l15 != 0) return;                                        // D(m = 4 r + g, n = l15): lanes 0, 16, 32, 48 hold n = 0
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int col = (vis ? c0 : hc0) + 4 * r + g;
        if (col >= ncol) continue;
        if (vis) {
            const double gr = s1[r] / a.N;
            const double d = a.lr * (a.mom * a.dvb[col] + gr);
            a.dvb[col] = d;
            a.vb[col] = a.vb[col] + d;
        } else {
            const double qn = a.damping * a.q[col] + (1.0 - a.damping) * s2[r];
            a.q[col] = qn;
            const double pen = a.cost * (qn - a.target);
            a.pen[col] = pen;
            double gr = s1[r] / a.N;
            gr = gr - pen;
            const double d = a.lr * (a.mom * a.dhb[col] + gr);
            a.dhb[col] = d;
            a.hb[col] = a.hb[col] + d;
        }
    }
}

__global__ __launch_bounds__(64) void bias_kernel(BiasArgs a) { bias_wave(a, blockIdx.x, threadIdx.x); }

// raw CD gradient + update of W (and of the transpose Wt) in one pass: thread (j = visible, i = hidden)
//   acc = sum_b h0m[b][i] X[b][j]  then  acc = fma(hm[b][i], -vs[b][j], acc)   (one chain)
struct GradArgs {
    const double *h0m, *hm, *X, *vs;     // [B][H], [B][H], [B][V] pitch ldx, [B][V]
    int ldx, B, V, H;
    double *W, *dW, *Wt;
    const double *pen;
    double N, l2, lr, mom;
    // sparsity_cost == 0 (the penalty the tiles read is zero whatever the bias update does): the bias / q_means
    // update rides in extra grid rows of the same launch, one wave per 16 columns, in the shadow of the tiles
    int ny, nbias;               // tile rows of the grid; 16-column bias jobs (0: separate bias_kernel launch)
    BiasArgs bias;
};
__global__ __launch_bounds__(256, 1) void grad_kernel(GradArgs a) {
    if ((int)blockIdx.y >= a.ny) {                       // block-uniform
        const int job = (((int)blockIdx.y - a.ny) * (int)gridDim.x + (int)blockIdx.x) * 4 + (int)(threadIdx.x >> 6);
        if (job < a.nbias) bias_wave(a.bias, job, threadIdx.x & 63);
        return;
    }
    // workgroup = 64 hidden (i) x 32 visible (j) of W; the chain runs over the rows b: positive phase, then the
    // negative phase with the visible operand negated (fma(h, -v, acc), as the oracle)
    __shared__ __attribute__((aligned(16))) double smem[2 * (T64_PBUF + T64_QBUF)];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wi = w & 1, wj = w >> 1, l15 = lane & 15, g = lane >> 4;
    const int i0 = blockIdx.x * T64_TI, j0 = blockIdx.y * T64_TJ;
    d4 acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    T64Seg sg[2] = {{a.h0m, a.H, a.X, a.ldx, a.B, 1.0}, {a.hm, a.H, a.vs, a.V, a.B, -1.0}};
    t64_mainloop<true>(acc, sg, 2, a.H, a.V, i0, j0, smem);
    const int j = j0 + wj * 16 + l15;
    if (j >= a.V) return;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + wi * 32 + 16 * t + 4 * r + g;
            if (i >= a.H) continue;
            const size_t e = (size_t)j * a.H + i;
            double gr = acc[t][r] / a.N;
            gr = gr - a.l2 * a.W[e];
            gr = gr - a.pen[i];
            const double d = a.lr * (a.mom * a.dW[e] + gr);
            a.dW[e] = d;
            const double wn = a.W[e] + d;
            a.W[e] = wn;
            a.Wt[(size_t)i * a.V + j] = wn;
        }
}

__global__ void prep_kernel(const double *X, double *Y, const double *sigma, int rows, int cols, double keep,
                            PhiloxKey key, unsigned long long flat0) {
    const size_t n = (size_t)rows * cols;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        double x = X[e];
        if (sigma) x = x / sigma[e % (size_t)cols];                       // rbm.py:107
        if (keep >= 0.0) x = (x / keep) * floor(keep + uniform_at(key, flat0 + e));   // tf.nn.dropout
        Y[e] = x;
    }
}
__global__ void transpose_kernel(const double *W, double *Wt, int V, int H) {
    const size_t n = (size_t)V * H;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
        Wt[(e % (size_t)H) * V + e / (size_t)H] = W[e];
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
// Reductions of the metrics are DETERMINISTIC (the fp32 path's row sums are): no atomics.  Every
// workgroup (256 threads) combines its four wave sums in wave order and stores ONE partial; reduce_fixed_kernel adds
// the partials of each quantity in a fixed order (strided per thread, xor tree, waves in order).
__device__ __forceinline__ void block_store(double v, double *dst, double (*s_w)[4], int slot) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_w[slot][threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) *dst = ((s_w[slot][0] + s_w[slot][1]) + s_w[slot][2]) + s_w[slot][3];
}
constexpr int SQ_BLOCKS = 128;
__global__ __launch_bounds__(256) void sqdiff_kernel(const double *A, const double *B, size_t n, double *part) {
    __shared__ double s_w[1][4];
    double s = 0.0;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const double d = B ? A[e] - B[e] : A[e];
        s += d * d;
    }
    block_store(s, part + blockIdx.x, s_w, 0);
}
// out[j] = sum of part[off[j] .. off[j] + cnt[j]) for up to 5 quantities, fixed order; cnt 0: out[j] = 0
struct ReduceJobs { int off[5], cnt[5]; };
__global__ __launch_bounds__(256) void reduce_fixed_kernel(const double *part, ReduceJobs jb, double *out) {
    __shared__ double s_w[5][4];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        double s = 0.0;
        for (int e = threadIdx.x; e < jb.cnt[j]; e += 256) s += part[jb.off[j] + e];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) s_w[j][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < 5) out[threadIdx.x] = ((s_w[threadIdx.x][0] + s_w[threadIdx.x][1]) + s_w[threadIdx.x][2]) + s_w[threadIdx.x][3];
}
__global__ void pll_index_kernel(int *out, int B, int V, PhiloxKey key, unsigned long long row0) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned long long idx = row0 + b;
    uint32_t w[4];
    bm::philox_block(key, idx >> 2, w);
    out[b] = (int)(w[idx & 3] % (uint32_t)V);
}
// MultinomialLayer in double (layers.py:54-70), the twin of bm::softmax_multinomial_kernel and of the oracle's
// softmax_multinomial_row_d: one wave per row of logits L[row][0..I) (written by act_kernel kind 3), in place:
//   mx = max l;  e[i] = exp_neg(min(mx - l[i], 700));  c[i] = c[i-1] + e[i] SEQUENTIALLY;  S = c[I-1];
//   means[i] = M * (e[i] / S);  draw d: t = u(row*M + d) * S, category = first c[i] > t.   LDS: c[I] | e[I] doubles
__global__ __launch_bounds__(64) void softmax_multinomial_kernel(double *L, int I, int M, int sample, double *states,
                                                                 PhiloxKey key, long long row0) {
    extern __shared__ double sm64[];
    double *c = sm64, *e = sm64 + I;
    const int row = blockIdx.x, lane = threadIdx.x;
    double *l = L + (size_t)row * I;
    double mx = -1.7976931348623157e308;
    for (int i = lane; i < I; i += 64) mx = fmax(mx, l[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    for (int i = lane; i < I; i += 64) {
        double d = mx - l[i];
        if (d > 700.0) d = 700.0;
        e[i] = exp_neg(d);
    }
    __syncthreads();
    if (lane == 0) {
        double run = 0.0;
        for (int i = 0; i < I; ++i) { run = run + e[i]; c[i] = run; }
    }
    __syncthreads();
    const double S = c[I - 1], Mf = (double)M;
    for (int i = lane; i < I; i += 64) {
        const double m = Mf * (e[i] / S);
        l[i] = m;
        if (states && !sample) states[(size_t)row * I + i] = m;
    }
    if (!states || !sample) return;
    __syncthreads();
    int *cnt = reinterpret_cast<int *>(e);
    for (int i = lane; i < I; i += 64) cnt[i] = 0;
    __syncthreads();
    for (int d = lane; d < M; d += 64) {
        const double u = uniform_at(key, (unsigned long long)(row0 + row) * (unsigned long long)M + (unsigned long long)d);
        const double t = u * S;
        int lo = 0, hi = I - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (c[mid] > t) hi = mid; else lo = mid + 1;
        }
        atomicAdd(cnt + lo, 1);
    }
    __syncthreads();
    for (int i = lane; i < I; i += 64) states[(size_t)row * I + i] = (double)cnt[i];
}
// h_hat ~ Multinomial(M, uniform over K) (rbm.py:58): counts of floor(u * K); three vectors (streams t = 0, 1, 2:
// free_energy_op, F(x) and F(x~) of the PLL); hhat [3][K] zeroed by the caller
__global__ void mn_hhat_kernel(double *hhat, int K, int M, PhiloxKey k0, PhiloxKey k1, PhiloxKey k2) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= M) return;
    const PhiloxKey keys[3] = {k0, k1, k2};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        int idx = (int)(uniform_at(keys[t], (unsigned long long)d) * (double)K);
        if (idx > K - 1) idx = K - 1;
        atomicAdd(hhat + (size_t)t * K + idx, 1.0);
    }
}
// MultinomialRBM free energy (rbm.py:52-62, without the lgamma constant): out[0] += F_hhat0(x), out[1] += F_hhat2(x~),
// out[2] += F_hhat1(x);  F_hhat(v) = -v.vb - sum_h (v W)_h * hhat_h
__global__ __launch_bounds__(256) void free_energy_mn_kernel(const double *X, int ldx, int B, int V, int H,
                                                             const double *W, const double *vb, const double *hhat,
                                                             const int *flip, double *rows /* [3][ldr] */, int ldr) {
    __shared__ double s_w[3][4];
    const int b = blockIdx.x;
    const double *x = X + (size_t)b * ldx;
    const int fc = flip ? flip[b] : -1;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        const double xv = x[v], xf = (v == fc) ? 1.0 - xv : xv;
        t0 -= xv * vb[v]; t1 -= xv * vb[v]; t2 -= xf * vb[v];
    }
    const double delta = (fc >= 0) ? 1.0 - 2.0 * x[fc] : 0.0;
    for (int h = threadIdx.x; h < H; h += blockDim.x) {
        double z = 0.0;
        for (int v = 0; v < V; ++v) z = fma(x[v], W[(size_t)v * H + h], z);
        t0 -= z * hhat[h];
        t1 -= z * hhat[(size_t)H + h];
        if (fc >= 0) t2 -= (z + delta * W[(size_t)fc * H + h]) * hhat[2 * (size_t)H + h];
    }
    // rows[0]: F(x), rows[1]: F(x~), rows[2]: the PLL's own F(x)
    block_store(t0, rows + b, s_w, 0);
    block_store(t2, rows + ldr + b, s_w, 1);
    block_store(t1, rows + 2 * (size_t)ldr + b, s_w, 2);
}
// free energy of rows (rbm.py:17-22 / :109-116), optionally of the row with column flip[b] flipped:
// out[0] += F(x_b), out[1] += F(x~_b).  One workgroup per row; hidden units strided over the threads.
__global__ __launch_bounds__(256) void free_energy_kernel(const double *X, int ldx, int B, int V, int H,
                                                          const double *W, const double *vb, const double *hb,
                                                          const double *sigma, const int *flip, double *rows /* [2][ldr] */, int ldr) {
    __shared__ double s_w[2][4];
    const int b = blockIdx.x;
    const double *x = X + (size_t)b * ldx;
    const int fc = flip ? flip[b] : -1;
    double t = 0.0, t2 = 0.0;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        const double xv = x[v], xf = (v == fc) ? 1.0 - xv : xv;
        if (sigma) {
            const double mu = vb[v] / sigma[v];
            t += 0.5 * (xv - mu) * (xv - mu);
            t2 += 0.5 * (xf - mu) * (xf - mu);
        } else {
            t -= xv * vb[v];
            t2 -= xf * vb[v];
        }
    }
    const double delta = (fc >= 0) ? 1.0 - 2.0 * x[fc] : 0.0;
    for (int h = threadIdx.x; h < H; h += blockDim.x) {
        double z = hb[h];
        for (int v = 0; v < V; ++v) z = fma(x[v], W[(size_t)v * H + h], z);
        t -= softplus(z);
        if (fc >= 0) t2 -= softplus(z + delta * W[(size_t)fc * H + h]);
    }
    block_store(t, rows + b, s_w, 0);
    block_store(t2, rows + ldr + b, s_w, 1);
}

struct DBuf {
    double *p = nullptr; size_t n = 0;
    int alloc(size_t count) {
        n = count;
        if (hipMalloc((void **)&p, (count ? count : 1) * sizeof(double)) != hipSuccess) { bm::set_error("hipMalloc of %zu doubles failed", count); return 1; }
        return hipMemset(p, 0, (count ? count : 1) * sizeof(double)) == hipSuccess ? 0 : 1;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; }
};
enum : uint32_t { SITE_DROPOUT = 1, SITE_H0 = 2, SITE_V = 3, SITE_H = 4, SITE_PLL = 5, SITE_FE = 6 };
}  // namespace bm64

struct bm_rbm64 {
    bm_rbm_config cfg;
    int V, H, maxB;
    hipStream_t stream = nullptr;
    bm64::DBuf W, Wt, dW, vb, hb, dvb, dhb, q, sigma, pen;
    bm64::DBuf h0m, h0s, hm, hs, vm, vs, Xp;
    int *flip = nullptr;
    double *scal = nullptr;      // [6] msre, l2, F(x), F(x~), F'(x) (multinomial: the PLL's own h_hat), spare
    bm64::DBuf hhat;             // [3*H] MultinomialRBM free-energy h_hat vectors (rbm.py:58)
    bm64::DBuf part;             // partial sums of the metric reductions: [2][SQ_BLOCKS] msre / l2 | [3][maxB] free-energy rows
    bool multinomial() const { return cfg.h_unit == BM_UNIT_MULTINOMIAL; }
    uint64_t seed = 0; uint32_t call = 0; int64_t row0 = 0;
    const double *Xin = nullptr; int Xin_ld = 0;
    // hyper-parameters as doubles (a Python float is a double; the float fields of cfg would round them)
    double l2, sp_target, sp_cost, sp_damping, dropout;
};

namespace bm64 {
static PhiloxKey make_key(const bm_rbm64 *h, uint32_t site, int t) {
    PhiloxKey k;
    k.k0 = (uint32_t)h->seed; k.k1 = (uint32_t)(h->seed >> 32);
    k.site = site + 16u * (uint32_t)t; k.call = h->call;
    return k;
}
static void launch_act(bm_rbm64 *h, bool up, const double *in, int ldin, int B, double *means, double *states,
                       int sample, uint32_t site, int t) {
    ActArgs a;
    memset(&a, 0, sizeof(a));
    if (up) { a.P = h->W.p; a.ldp = h->H; a.K = h->V; a.I = h->H; a.bias = h->hb.p; a.kind = BM_UNIT_BERNOULLI;
              a.mult = 1.0 + (h->cfg.dbm_first ? 1.0 : 0.0); }
    else    { a.P = h->Wt.p; a.ldp = h->V; a.K = h->H; a.I = h->V; a.bias = h->vb.p; a.sigma = h->sigma.p; a.kind = h->cfg.v_unit;
              a.mult = 1.0 + (h->cfg.dbm_last ? 1.0 : 0.0); }
    a.Q = in; a.ldq = ldin; a.J = B; a.sample = sample; a.means = means; a.states = states;
    a.key = make_key(h, site, t); a.row0 = h->row0;
    if (up && h->multinomial()) {
        // MultinomialLayer (layers.py:54-70): logits from the GEMM, then one wave per row for the softmax
        // (activation) and the multinomial counts (sample)
        a.kind = 3; a.sample = 0; a.states = nullptr;
        hipLaunchKernelGGL(act_kernel, dim3((a.I + T64_TI - 1) / T64_TI, (B + T64_TJ - 1) / T64_TJ), dim3(256), 0, h->stream, a);
        hipLaunchKernelGGL(softmax_multinomial_kernel, dim3(B), dim3(64), 2 * (size_t)h->H * sizeof(double), h->stream,
                           means, h->H, h->cfg.n_samples, sample, states, a.key, (long long)h->row0);
        return;
    }
    hipLaunchKernelGGL(act_kernel, dim3((a.I + T64_TI - 1) / T64_TI, (B + T64_TJ - 1) / T64_TJ), dim3(256), 0, h->stream, a);
}
// -lgamma(M + K) + lgamma(M + 1) + lgamma(K)  (rbm.py:61); 0 for the other RBMs
static double mn_fe_const(const bm_rbm64 *h) {
    if (!h->multinomial()) return 0.0;
    const double M = h->cfg.n_samples, K = h->H;
    return -lgamma(M + K) + lgamma(M + 1.0) + lgamma(K);
}
// free energies of the rows of Xin into scal[2] (F(x)), scal[3] (F(x~), with flip) and, MultinomialRBM, scal[4]
static void launch_fe(bm_rbm64 *h, const double *Xin, int ldx, int B, const int *flip) {
    if (h->multinomial()) {                    // rbm.py:52-62: fresh h_hat draws, streams t = 0, 1, 2
        (void)hipMemsetAsync(h->hhat.p, 0, 3 * (size_t)h->H * sizeof(double), h->stream);
        const int M = h->cfg.n_samples;
        hipLaunchKernelGGL(mn_hhat_kernel, dim3((M + 255) / 256), dim3(256), 0, h->stream, h->hhat.p, h->H, M,
                           make_key(h, SITE_FE, 0), make_key(h, SITE_FE, 1), make_key(h, SITE_FE, 2));
        hipLaunchKernelGGL(free_energy_mn_kernel, dim3(B), dim3(256), 0, h->stream, Xin, ldx, B, h->V, h->H,
                           (const double *)h->W.p, (const double *)h->vb.p, (const double *)h->hhat.p, flip,
                           h->part.p + 2 * SQ_BLOCKS, h->maxB);
        return;
    }
    hipLaunchKernelGGL(free_energy_kernel, dim3(B), dim3(256), 0, h->stream, Xin, ldx, B, h->V, h->H,
                       (const double *)h->W.p, (const double *)h->vb.p, (const double *)h->hb.p,
                       (const double *)(h->cfg.v_unit == BM_UNIT_GAUSSIAN ? h->sigma.p : nullptr), flip,
                       h->part.p + 2 * SQ_BLOCKS, h->maxB);
}
// scal[0..4] = msre sum, l2 sum, sum F(x), sum F(x~), sum F'(x) from the partials, in a fixed order
static void launch_reduce(bm_rbm64 *h, int B, bool with_sq, bool with_flip) {
    ReduceJobs jb;
    const int fe0 = 2 * SQ_BLOCKS;
    jb.off[0] = 0;          jb.cnt[0] = with_sq ? SQ_BLOCKS : 0;
    jb.off[1] = SQ_BLOCKS;  jb.cnt[1] = with_sq ? SQ_BLOCKS : 0;
    jb.off[2] = fe0;                 jb.cnt[2] = B;
    jb.off[3] = fe0 + h->maxB;       jb.cnt[3] = with_flip ? B : 0;
    jb.off[4] = fe0 + 2 * h->maxB;   jb.cnt[4] = h->multinomial() ? B : 0;
    hipLaunchKernelGGL(reduce_fixed_kernel, dim3(1), dim3(256), 0, h->stream, (const double *)h->part.p, jb, h->scal);
}
// input preprocessing + h0 + k Gibbs steps (base_rbm.py:417-426)
static int run_chain(bm_rbm64 *h, const double *X_dev, int B, int k, double *hm_out) {
    BM_CHECK(B >= 1 && B <= h->maxB, "batch %d outside [1, max_batch=%d]", B, h->maxB);
    BM_CHECK(k >= 1, "n_gibbs_steps must be >= 1 (got %d)", k);
    const double *Xin = X_dev;
    if (h->cfg.v_unit == BM_UNIT_GAUSSIAN || h->dropout >= 0.0) {
        hipLaunchKernelGGL(prep_kernel, dim3(256), dim3(256), 0, h->stream, X_dev, h->Xp.p,
                           (const double *)(h->cfg.v_unit == BM_UNIT_GAUSSIAN ? h->sigma.p : nullptr), B, h->V,
                           h->dropout >= 0.0 ? h->dropout : -1.0, make_key(h, SITE_DROPOUT, 0),
                           (unsigned long long)h->row0 * (unsigned long long)h->V);
        Xin = h->Xp.p;
    }
    h->Xin = Xin; h->Xin_ld = h->V;
    launch_act(h, true, Xin, h->V, B, h->h0m.p, h->h0s.p, 1, SITE_H0, 0);
    const double *hstate = h->cfg.sample_h_states ? h->h0s.p : h->h0m.p;
    for (int t = 0; t < k; ++t) {
        launch_act(h, false, hstate, h->H, B, h->vm.p, h->vs.p, h->cfg.sample_v_states, SITE_V, t);
        launch_act(h, true, h->vs.p, h->V, B, (hm_out && t == k - 1) ? hm_out : h->hm.p, h->hs.p,
                   h->cfg.sample_h_states, SITE_H, t);
        hstate = h->hs.p;
    }
    return 0;
}
static void launch_update(bm_rbm64 *h, int B, double lr, double mom) {
    BiasArgs b;
    b.X = h->Xin; b.vs = h->vs.p; b.h0m = h->h0m.p; b.hm = h->hm.p; b.ldx = h->Xin_ld; b.B = B; b.V = h->V; b.H = h->H;
    b.vb = h->vb.p; b.dvb = h->dvb.p; b.hb = h->hb.p; b.dhb = h->dhb.p; b.q = h->q.p; b.pen = h->pen.p;
    b.N = (double)B; b.lr = lr; b.mom = mom;
    b.damping = h->sp_damping; b.cost = h->sp_cost; b.target = h->sp_target;
    const int nbias = (h->V + 15) / 16 + (h->H + 15) / 16;
    const bool fused = h->sp_cost == 0.0;
    if (!fused) hipLaunchKernelGGL(bias_kernel, dim3(nbias), dim3(64), 0, h->stream, b);
    GradArgs g;
    g.h0m = h->h0m.p; g.hm = h->hm.p; g.X = h->Xin; g.vs = h->vs.p; g.ldx = h->Xin_ld; g.B = B; g.V = h->V; g.H = h->H;
    g.W = h->W.p; g.dW = h->dW.p; g.Wt = h->Wt.p; g.pen = h->pen.p;
    g.N = (double)B; g.l2 = h->l2; g.lr = lr; g.mom = mom;
    const int nx = (h->H + T64_TI - 1) / T64_TI, ny = (h->V + T64_TJ - 1) / T64_TJ;
    g.ny = ny; g.nbias = fused ? nbias : 0; g.bias = b;
    const int extra = fused ? (nbias + 4 * nx - 1) / (4 * nx) : 0;
    hipLaunchKernelGGL(grad_kernel, dim3(nx, ny + extra), dim3(256), 0, h->stream, g);
}
static int metrics_from_chain(bm_rbm64 *h, int B, double *out4) {
    BM_HIP(hipMemsetAsync(h->scal, 0, 6 * sizeof(double), h->stream));
    hipLaunchKernelGGL(sqdiff_kernel, dim3(SQ_BLOCKS), dim3(256), 0, h->stream, h->Xin, (const double *)h->vm.p, (size_t)B * h->V, h->part.p);
    hipLaunchKernelGGL(sqdiff_kernel, dim3(SQ_BLOCKS), dim3(256), 0, h->stream, (const double *)h->W.p, (const double *)nullptr, (size_t)h->V * h->H, h->part.p + SQ_BLOCKS);
    hipLaunchKernelGGL(pll_index_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, h->flip, B, h->V,
                       make_key(h, SITE_PLL, 0), (unsigned long long)h->row0);
    launch_fe(h, h->Xin, h->Xin_ld, B, (const int *)h->flip);
    launch_reduce(h, B, true, true);
    double host[6];
    BM_HIP(hipMemcpyAsync(host, h->scal, sizeof(host), hipMemcpyDeviceToHost, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
    // MultinomialRBM: every _free_energy() call draws its own h_hat (host[4] = F(x) of the PLL pair) and carries the
    // constant of rbm.py:61 (it cancels in the PLL difference)
    const double fe = host[2] / B + mn_fe_const(h), fe1 = h->multinomial() ? host[4] / B + mn_fe_const(h) : fe;
    const double fe2 = host[3] / B + mn_fe_const(h), d = fe2 - fe1;
    out4[0] = host[0] / ((double)B * h->V);
    out4[1] = (double)h->V * -(fmax(-d, 0.0) + log1p(exp(-fabs(d))));
    out4[2] = h->l2 * (0.5 * host[1]);
    out4[3] = fe;
    return 0;
}
static DBuf *find(bm_rbm64 *h, const char *name) {
    struct { const char *n; DBuf *b; } t[] = {{"W", &h->W}, {"vb", &h->vb}, {"hb", &h->hb}, {"dW", &h->dW}, {"dvb", &h->dvb},
                                              {"dhb", &h->dhb}, {"q_means", &h->q}, {"sigma", &h->sigma}};
    for (auto &e : t) if (!strcmp(e.n, name)) return e.b;
    return nullptr;
}
}  // namespace bm64

using namespace bm64;

extern "C" {

int bm_rbm64_create(const bm_rbm_config *cfg, const double *hyper5, bm_rbm64 **out) {
    BM_CHECK(cfg && out, "null argument");
    BM_CHECK(cfg->n_visible >= 1 && cfg->n_hidden >= 1, "bad layer sizes %d x %d", cfg->n_visible, cfg->n_hidden);
    BM_CHECK(cfg->max_batch >= 1, "max_batch must be >= 1");
    BM_CHECK(cfg->v_unit == BM_UNIT_BERNOULLI || cfg->v_unit == BM_UNIT_GAUSSIAN, "unknown visible unit %d", cfg->v_unit);
    BM_CHECK(cfg->h_unit == BM_UNIT_BERNOULLI || cfg->h_unit == BM_UNIT_MULTINOMIAL, "unknown hidden unit %d", cfg->h_unit);
    BM_CHECK(cfg->h_unit != BM_UNIT_MULTINOMIAL || (cfg->n_samples >= 1 && cfg->n_hidden <= 8192),
             "MultinomialRBM needs n_samples >= 1 and n_hidden <= 8192 (got %d, %d)", cfg->n_samples, cfg->n_hidden);
    BM_CHECK(bm_device_count() > 0, "no HIP device visible: libbm355 has no CPU fallback");
    bm_rbm64 *h = new bm_rbm64();
    h->cfg = *cfg;
    h->V = cfg->n_visible; h->H = cfg->n_hidden; h->maxB = cfg->max_batch;
    h->l2 = hyper5 ? hyper5[0] : (double)cfg->l2;
    h->sp_target = hyper5 ? hyper5[1] : (double)cfg->sparsity_target;
    h->sp_cost = hyper5 ? hyper5[2] : (double)cfg->sparsity_cost;
    h->sp_damping = hyper5 ? hyper5[3] : (double)cfg->sparsity_damping;
    h->dropout = hyper5 ? hyper5[4] : (double)cfg->dropout;
    const size_t V = h->V, H = h->H, B = h->maxB;
    BM_HIP(hipStreamCreate(&h->stream));
    BM_TRY(h->W.alloc(V * H)); BM_TRY(h->Wt.alloc(V * H)); BM_TRY(h->dW.alloc(V * H));
    BM_TRY(h->vb.alloc(V)); BM_TRY(h->hb.alloc(H)); BM_TRY(h->dvb.alloc(V)); BM_TRY(h->dhb.alloc(H));
    BM_TRY(h->q.alloc(H)); BM_TRY(h->sigma.alloc(V)); BM_TRY(h->pen.alloc(H));
    BM_TRY(h->h0m.alloc(B * H)); BM_TRY(h->h0s.alloc(B * H)); BM_TRY(h->hm.alloc(B * H)); BM_TRY(h->hs.alloc(B * H));
    BM_TRY(h->vm.alloc(B * V)); BM_TRY(h->vs.alloc(B * V)); BM_TRY(h->Xp.alloc(B * V));
    BM_HIP(hipMalloc((void **)&h->flip, B * sizeof(int)));
    BM_HIP(hipMalloc((void **)&h->scal, 6 * sizeof(double)));
    BM_TRY(h->hhat.alloc(3 * H));
    BM_TRY(h->part.alloc(2 * (size_t)bm64::SQ_BLOCKS + 3 * (size_t)h->maxB));
    std::vector<double> ones(V, 1.0);
    BM_HIP(hipMemcpy(h->sigma.p, ones.data(), V * sizeof(double), hipMemcpyHostToDevice));
    *out = h;
    return 0;
}

int bm_rbm64_destroy(bm_rbm64 *h) {
    if (!h) return 0;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    DBuf *all[] = {&h->W, &h->Wt, &h->dW, &h->vb, &h->hb, &h->dvb, &h->dhb, &h->q, &h->sigma, &h->pen,
                   &h->h0m, &h->h0s, &h->hm, &h->hs, &h->vm, &h->vs, &h->Xp};
    for (DBuf *b : all) b->release();
    if (h->flip) (void)hipFree(h->flip);
    if (h->scal) (void)hipFree(h->scal);
    h->hhat.release();
    h->part.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

int bm_rbm64_sync(bm_rbm64 *h) { BM_HIP(hipStreamSynchronize(h->stream)); return 0; }
int bm_rbm64_seed(bm_rbm64 *h, uint64_t seed) { h->seed = seed; h->call = 0; return 0; }
int bm_rbm64_set_row_offset(bm_rbm64 *h, int64_t row0) { h->row0 = row0; return 0; }

int bm_rbm64_set_param(bm_rbm64 *h, const char *name, const double *host, size_t n) {
    DBuf *b = find(h, name);
    BM_CHECK(b, "unknown variable '%s'", name);
    BM_CHECK(n == b->n, "variable '%s' has %zu elements, got %zu", name, b->n, n);
    BM_HIP(hipStreamSynchronize(h->stream));
    BM_HIP(hipMemcpy(b->p, host, n * sizeof(double), hipMemcpyHostToDevice));
    if (b == &h->W) {
        hipLaunchKernelGGL(transpose_kernel, dim3(256), dim3(256), 0, h->stream, (const double *)h->W.p, h->Wt.p, h->V, h->H);
        BM_HIP(hipStreamSynchronize(h->stream));
    }
    return 0;
}
int bm_rbm64_get_param(bm_rbm64 *h, const char *name, double *host, size_t n) {
    DBuf *b = find(h, name);
    BM_CHECK(b, "unknown variable '%s'", name);
    BM_CHECK(n == b->n, "variable '%s' has %zu elements, got %zu", name, b->n, n);
    BM_HIP(hipStreamSynchronize(h->stream));
    BM_HIP(hipMemcpy(host, b->p, n * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int bm_rbm64_train_step(bm_rbm64 *h, const double *X_dev, int32_t B, double lr, double mom, int32_t k) {
    BM_TRY(run_chain(h, X_dev, B, k, nullptr));
    launch_update(h, B, lr, mom);
    h->call++;
    BM_HIP(hipGetLastError());
    return 0;
}
int bm_rbm64_train_step_metrics(bm_rbm64 *h, const double *X_dev, int32_t B, double lr, double mom, int32_t k, double *out4) {
    BM_TRY(run_chain(h, X_dev, B, k, nullptr));
    BM_TRY(metrics_from_chain(h, B, out4));
    launch_update(h, B, lr, mom);
    h->call++;
    BM_HIP(hipGetLastError());
    return 0;
}
int bm_rbm64_transform(bm_rbm64 *h, const double *X_dev, int32_t B, int32_t k, double *H_dev) {
    BM_CHECK(H_dev, "null output");
    BM_TRY(run_chain(h, X_dev, B, k, H_dev));
    h->call++;
    BM_HIP(hipGetLastError());
    return 0;
}
int bm_rbm64_metrics(bm_rbm64 *h, const double *X_dev, int32_t B, int32_t k, double *out4) {
    BM_TRY(run_chain(h, X_dev, B, k, nullptr));
    BM_TRY(metrics_from_chain(h, B, out4));
    h->call++;
    return 0;
}
int bm_rbm64_free_energy(bm_rbm64 *h, const double *X_dev, int32_t B, double *out1) {
    BM_CHECK(B >= 1 && B <= h->maxB, "batch %d outside [1, max_batch=%d]", B, h->maxB);
    const double *Xin = X_dev;
    const bool dropped = h->dropout >= 0.0;      // free_energy_op sees the dropped input (base_rbm.py:417-418, :516)
    if (h->cfg.v_unit == BM_UNIT_GAUSSIAN || dropped) {
        hipLaunchKernelGGL(prep_kernel, dim3(256), dim3(256), 0, h->stream, X_dev, h->Xp.p,
                           (const double *)(h->cfg.v_unit == BM_UNIT_GAUSSIAN ? h->sigma.p : nullptr), B, h->V,
                           dropped ? h->dropout : -1.0, make_key(h, bm64::SITE_DROPOUT, 0),
                           (unsigned long long)h->row0 * (unsigned long long)h->V);
        Xin = h->Xp.p;
    }
    BM_HIP(hipMemsetAsync(h->scal, 0, 6 * sizeof(double), h->stream));
    launch_fe(h, Xin, h->V, B, (const int *)nullptr);
    bm64::launch_reduce(h, B, false, false);
    double host[6];
    BM_HIP(hipMemcpyAsync(host, h->scal, sizeof(host), hipMemcpyDeviceToHost, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
    *out1 = host[2] / B + mn_fe_const(h);
    if (dropped || h->multinomial()) h->call++;  // the dropout mask / the random h_hat consumed one call of the stream
    return 0;
}

}  // extern "C"
