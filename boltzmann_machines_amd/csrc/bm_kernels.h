// bm_kernels.h — the fused CDNA4 kernels of the Boltzmann-machine hot path.
//
//   act_kernel      K1/K2/K9/K10/K16 of SURVEY §2b: propagation GEMM (one or two
//                   K segments) + multiplier + bias + sigmoid / Gaussian-linear +
//                   Philox Bernoulli / Normal draw, all in the MFMA epilogue.
//   grad_kernel     K3/K13: positive and negative outer products as two MFMA
//                   chains + L2 + sparsity + momentum + in-place W update.
//   colstat_kernel  K4: column sums as an MFMA against a vector of ones (keeps the
//                   canonical sequential order) + bias / q_means updates.
//   elementwise / reduction helpers for dropout, msre, l2, free energy, PLL.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <array>
#include <map>
#include <mutex>
#include "bm_gemm.h"
#include "bm_bf3.h"
#include "bm_numerics.h"
#include "bm_rng.h"

namespace bm {

// device-side mean-field loop control (dbm.py:449-452), see mf_ctl_kernel
struct MfCtl { unsigned maxdiff; int done; int steps; float resid; };

// ------------------------------------------------------------------ act_kernel
struct ActArgs {
    Operand P1, Q1; int K1;      // segment 1:  z += sum_k P1[i][k] Q1[j][k]
    int p_xm;                    // 1: P1 is stored x-major [i][k] (W itself for the prop-down: no transpose kept);
                                 //    single segment, geometries with MI == 1
    Operand P2, Q2; int K2;      // segment 2 (K2 == 0: absent), chained onto the same accumulator
    int I, J;                    // output is [J rows][I cols], ld = ldo
    const float *bias;           // [I]
    const float *sigma;          // [I], Gaussian units only
    float mult;                  // propup/propdown multiplier or AIS beta applied to z
    float bmult;                 // multiplier applied to the bias (== mult except mean-field init, dbm.py:434-446)
    int kind;                    // 0 BERNOULLI: sigmoid(mult*z + mult*b); 1 GAUSSIAN: (mult*z)*sigma + mult*b;
                                 // 2: raw mult*z; 3: logits mult*z + bmult*b (input of softmax_multinomial_kernel)
    int sample;                  // 1: states = draw(means); 0: states = means
    int lit;                     // 1: Bernoulli units use sigmoid_literal (bm_numerics.h; bm_dbm_set_sigmoid_literal) - a launch
                                 //    flavour of its own (launch_act_lit), never a branch of the default kernels
    float *means;                // may be null
    float *states;               // may be null
    float *negmeans;             // may be null: -means (P operand of the negative phase in grad_kernel form 0)
    int ldo;
    PhiloxKey key;
    long long row0;              // global row of local row 0 (rank-invariant bitmaps)
    const float *prev;           // mean-field: previous mu (same layout as means) or null
    unsigned *maxdiff;           // mean-field: atomicMax target for ||mu_new - mu||_inf (float bits)
    float *maxdiff_blk;          // or (launches of <= BM_MF_SLOTS workgroups): one slot per workgroup, reduced by
                                 // mf_ctl_kernel - hundreds of same-address atomics cost ~4 us per sweep kernel
    // optional per-row reductions of the epilogue (AIS log-weights dbm.py:650-660,713-720; ELBO :741-745).
    // DETERMINISTIC: no atomics.  Every aligned group of 16 output columns ("slot" s = i / 16) of row j
    // gets ONE partial sum, computed in a fixed order that does not depend on the tile geometry
    // (quads of 4 consecutive i summed left to right, then (q0+q1)+(q2+q3)), stored at
    // part[s * ld_part + j]; the consumer kernel (ais_score_kernel / elbo_row_kernel) adds the
    // ceil(I/16) slots of a row in ascending order.  Round 1 used fp32 atomicAdd from every tile:
    // order - and therefore the low bits of log Z - varied from run to run.
    float *rowacc;               // [ceil(I/16)][ld_part]: sum_i softplus(beta_b*(z+b)) - softplus(beta_a*(z+b))  (AIS)
                                 //     or sum_i z * dot_mat[j][i]                                                 (ELBO, dot_mat set)
    float beta_a, beta_b;
    int rowacc_single;           // 1: rowacc = sum_i softplus(beta_b*(z+b)) alone (literal fp32 AIS, bm_dbm_set_ais_literal)
    float *rowdot_out;           // [ceil(I/16)][ld_part]: sum_i states[j][i] * dot_vec[i]
    int ld_part;                 // pitch of the partial-sum buffers (>= J)
    const float *dot_vec;        // [I]
    const float *dot_mat;        // [J][I] pitch ld_dot
    int ld_dot;
    // mean-field plumbing: a loop-invariant partial pre-activation to start the chain from
    // (X.W0, hoisted out of the sweeps: continuing the chain from it is bit-identical to
    // recomputing segment 1), and a device-side "loop finished" flag that turns the launch into a no-op
    const float *acc_init;       // [J][I] pitch ld_init or null
    int ld_init;
    const int *skip;             // device int: != 0 -> return immediately
    // mean-field, first kernel of sweep s: the loop-control update for sweep s-1 ("Check(s-1)": steps += 1,
    // done = !(residual > tol)) is evaluated HERE by every workgroup from the residual slots sweep s-1 left
    // (workgroup 0 commits it to chk_ctl), instead of by a one-workgroup kernel between the sweeps: one kernel
    // boundary less per sweep (~3.5 us of ~25 at the 784-512-1024 shape).  Null: plain `skip` behaviour.
    MfCtl *chk_ctl;
    const float *chk_slots; int chk_n; float chk_tol;
    // fast-binary mode (bm_bf3.h): the contraction from bf16 weight planes and bf16 state shadows (b3.K1 > 0), and
    // the bf16 shadow of the states this launch writes (null: none)
    Bf3Range b3;
    uint16_t *states16; int ld16;
    int geo_hint;                // != 0: this tile geometry (launch_act_as codes) instead of the launch tuner's choice - the caller
                                 // knows something the tuner's solo timing does not (the DBM's particle sweeps run BESIDE the
                                 // mean-field passes: the 32 KiB tile shares a CU with them, the tuner's 96 KiB pick takes turns)
    int map_xi;                  // block -> tile map: 0 = the traffic model's XCD grid, 8 / 4 / 2 / 1 that grid, -1 the slab order (launch tuner)
    // Metric fetch of a training iteration (base_rbm.py:496-517, rbm.py:17-22): the h0 pass already holds the pre-activations
    // x.W + hb the free energy needs, so ITS epilogue leaves, per row j, sum_i softplus(z + b) (through `rowacc`) and the same for
    // the PLL partner x~ (one flipped column fc = fe_flip[j]: z~ = z + (1 - 2 x[j][fc]) W[fc][:], a rank-1 correction; through
    // fe_rowacc2) - what fe_hidden_kernel computes from a GEMM of its own (15 us at 784 x 1024 x 512).  Through the slot
    // partials of `rowacc` (rowacc_single = 1, beta_b = 1: sum_i softplus(z + b) per 16-column slot, no atomics, any tile
    // geometry the same bits); fe_rowacc2 is a second partial array [ceil(I/16)][ld_part] for the partner.  Null: off
    // (every other launch).  Needs mult == 1.
    float *fe_rowacc2;
    PhiloxKey fe_key;                    // fe_flip == FE_FLIP_FROM_KEY: the flip column is pll_flip_col(fe_key, row0 + j, K1) (no prep launch)
    double *fe_zero;                     // six accumulators this pass zeroes for the kernels of the fetch that follow it (or null)
    int fe_rm;                           // pitch of the two partial arrays when fe_flip is set: they are ROW major ([J][fe_rm]: the
                                         // row kernel reads a row's slots as one line), not slot major like the AIS partials
    const int *fe_flip;
    const float *fe_x; int fe_ldx;       // the pass's own input rows [J][K] (for x[j][fc])
    const float *fe_w; int fe_ldw;       // W [K][I] row-major (row fc)
#ifdef BM_PROBE
    long long *dbg;              // [grid][4] s_memtime stamps (tools/probe_act.hip only)
#endif
};
#ifdef BM_PROBE
#define BM_STAMP(n) do { if (a.dbg && threadIdx.x == 0) a.dbg[blockIdx.x * 4 + (n)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define BM_STAMP(n) do {} while (0)
#endif

// pll_rand = tf.random_uniform([B], 0, V, int32) (base_rbm.py:500-501): the column flipped in row `idx` (global row)
__device__ __forceinline__ int pll_flip_col(const PhiloxKey &key, unsigned long long idx, int V) {
    uint32_t w[4];
    philox_block(key, idx >> 2, w);
    return (int)(w[idx & 3] % (uint32_t)V);
}
#define FE_FLIP_FROM_KEY (reinterpret_cast<const int *>(uintptr_t(1)))
constexpr int BM_MF_SLOTS = 1024;      // per-workgroup residual slots per layer (ActArgs::maxdiff_blk)

// draw for 4 consecutive outputs starting at flat index `flat` (multiple of 4 on the fast path)
__device__ __forceinline__ void draw4(const ActArgs &a, const PhiloxKey &key, unsigned long long flat, int ib, int nvalid,
                                      const float *m, float *s, bool aligned) {
    if (aligned) {                  // the 4 outputs are exactly one Philox block
        uint32_t wds[4];
        philox_block(key, flat >> 2, wds);
        if (a.kind == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] = (u32_to_uniform(wds[r]) < m[r]) ? 1.f : 0.f;
        } else {
            float n[4];
            box_muller(wds[0], wds[1], n[0], n[1]);
            box_muller(wds[2], wds[3], n[2], n[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] = n[r] * a.sigma[ib + r] + m[r];
        }
    } else {                        // generic path (I % 4 != 0 or ragged edge): per element
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r >= nvalid) break;
            const unsigned long long idx = flat + r;
            uint32_t wds[4];
            philox_block(key, idx >> 2, wds);
            if (a.kind == 0) {
                s[r] = (u32_to_uniform(wds[idx & 3]) < m[r]) ? 1.f : 0.f;
            } else {
                float n0, n1;
                const int pr = (int)(idx & 3) >> 1;
                box_muller(wds[2 * pr], wds[2 * pr + 1], n0, n1);
                s[r] = ((idx & 1) ? n1 : n0) * a.sigma[ib + r] + m[r];
            }
        }
    }
}

// Output stores of the hot kernels (BM_NT_STORES, default on): streaming ("nt") stores.  Results
// are consumed by the NEXT kernel, after the kernel-boundary L2 writeback + invalidate of the 8
// non-coherent XCD L2s, so caching them write-back only queues them up for that flush.
#ifndef BM_NT_STORES
#define BM_NT_STORES 1
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void stream_store4(float *p, float x, float y, float z, float w) {
    if (BM_NT_STORES) __builtin_nontemporal_store((f32x4){x, y, z, w}, reinterpret_cast<f32x4 *>(p));
    else *reinterpret_cast<float4 *>(p) = make_float4(x, y, z, w);
}
__device__ __forceinline__ void stream_store2(float *p, float x, float y) {
    if (BM_NT_STORES) __builtin_nontemporal_store((f32x2){x, y}, reinterpret_cast<f32x2 *>(p));
    else *reinterpret_cast<float2 *>(p) = make_float2(x, y);
}

template <bool CACHED = false>
__device__ __forceinline__ void store4(float *dst, size_t o, const float *v, int nvalid, bool v4) {
    if (v4) {
        // CACHED: plain write-back stores (a lane of the MI = 2 tile writes its 32 bytes as two 16-byte stores 16 bytes
        // apart: as streaming stores each instruction leaves half-filled lines behind, WRITE_SIZE 1.8x the payload)
        if (CACHED) *reinterpret_cast<float4 *>(dst + o) = make_float4(v[0], v[1], v[2], v[3]);
        else stream_store4(dst + o, v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (r < nvalid) dst[o + r] = v[r];
    }
}

template <int MI> struct PhiloxFor { typedef PhiloxPair type; };
template <> struct PhiloxFor<1> { typedef PhiloxOne type; };

// max over the 64 lanes of a wave, valid in LANE 63: six DPP steps on the vector ALU (quad swaps, row rotations, the two row
// broadcasts of gfx9) instead of six ds_bpermute round trips through the LDS crossbar (__shfl_xor: ~100 cycles each, dependent)
__device__ __forceinline__ float wave_max_lane63(float v) {
#define BM_DPP_MAX(ctrl, rmask) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rmask, 0xf, false)))
    BM_DPP_MAX(0xb1, 0xf);     // quad_perm:[1,0,3,2]
    BM_DPP_MAX(0x4e, 0xf);     // quad_perm:[2,3,0,1]
    BM_DPP_MAX(0x124, 0xf);    // row_ror:4
    BM_DPP_MAX(0x128, 0xf);    // row_ror:8
    BM_DPP_MAX(0x142, 0xa);    // row_bcast:15 into rows 1 and 3
    BM_DPP_MAX(0x143, 0xc);    // row_bcast:31 into rows 2 and 3
#undef BM_DPP_MAX
    return v;
}

// act_kernel's side work for the main loop's pipeline fill.
// MF = the mean-field flavour (act_kernel<..., MF = true>; launch_act_mf): a pass of the data-dependent mean-field loop
// (dbm.py:429-478) carries plumbing no other pass has - the previous mu of the lane's outputs (residual), the loop control of
// the previous sweep (ActArgs::chk_ctl) or the loop's `done` word (ActArgs::skip), a stored partial pre-activation
// (ActArgs::acc_init).  tools/probe_mf.hip priced it at 2.5 of 12.7 us (h1 pass) and 1.3 of 10.7 us (h2 pass) at 784-512-1024
// x 512 while all of it sat on the critical path: the control check was one global round trip + a reduction BEFORE the first
// operand load went out, and the epilogue inputs were loaded behind the LDS-DMA pieces, where the counted wait of the fill
// (vmcnt(pieces still allowed in flight)) then had to retire a third chunk and the loads themselves.  Here:
//   * preload(): every epilogue input is requested BEFORE the first DMA piece (16-byte loads where the run is aligned), so
//     the fill's counted wait is exact again;
//   * the control check's loads go out BEHIND the DMA pieces (fill()): they return when the fill has landed, the wave
//     partials meet at the barrier the fill ends with anyway, and post_fill() decides; a finished loop costs a launch and a
//     fill, not a K loop.
template <int E, class Rng, bool MF = false, int NTH = 256> struct ActSide {
    static constexpr bool kFinalSync = false;    // one pipeline per kernel: waves enter the epilogue as they finish
    static constexpr bool kSplitFill = false, kCohQ = false, kCanAbort = MF;
    static constexpr bool kMF = MF;
    const float *bias, *sigma;
    const float *prev_row;       // mean-field: &prev[j][ib0] when the row is valid, else null
    int ib0, I, with_rng;
    float bs[E], sg[E], pv[E];   // pv: previous mu of the lane's outputs (mean-field residual)
    Rng rng;
    // MF only
    MfCtl *chk_ctl; const float *chk_slots; int chk_n; float chk_tol; const int *skip; float *s_chk; int aborted, nthreads;

    __device__ __forceinline__ void load_inputs() {
        const float *sp = sigma ? sigma : bias;     // unconditional loads + select: no branch, no early wait
        const bool has_sigma = sigma != nullptr;
        if (MF && prev_row) {                       // wave-uniform per kernel (null for all lanes or row-dependent)
            if (E == 4 && ib0 + 3 < I && (((uintptr_t)prev_row & 15u) == 0)) {
                const float4 t = *reinterpret_cast<const float4 *>(prev_row);
                pv[0] = t.x; pv[1] = t.y; pv[2] = t.z; pv[E - 1] = t.w;
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) pv[e] = (ib0 + e < I) ? prev_row[e] : 0.f;
            }
        }
        if (MF && E == 4 && ib0 + 3 < I && (((uintptr_t)(bias + ib0) & 15u) == 0) && (((uintptr_t)(sp + ib0) & 15u) == 0)) {
            const float4 b = *reinterpret_cast<const float4 *>(bias + ib0), v = *reinterpret_cast<const float4 *>(sp + ib0);
            bs[0] = b.x; bs[1] = b.y; bs[2] = b.z; bs[E - 1] = b.w;
            sg[0] = has_sigma ? v.x : 1.0f; sg[1] = has_sigma ? v.y : 1.0f; sg[2] = has_sigma ? v.z : 1.0f; sg[E - 1] = has_sigma ? v.w : 1.0f;
            return;
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = (ib0 + e < I) ? ib0 + e : I - 1;
            bs[e] = bias[i];
            const float sv = sp[i];
            sg[e] = has_sigma ? sv : 1.0f;
        }
    }
    // The control words of the loop (the residual slots of the previous sweep, the `done` word) travel like the operand
    // chunks: by LDS-DMA, requested in preload() - OLDER than every DMA piece of the fill, so the fill's own counted wait and
    // barrier retire them - and are read from LDS in post_fill().  (A load into registers would get hipcc's
    // `s_waitcnt vmcnt(0)` in front of its first use, which also waits for the youngest chunk in flight: +0.6 us per pass.)
    static constexpr int CHK_MAX = 4096;         // MAXL * BM_MF_SLOTS
    float *s_slots;                              // [CHK_MAX + 4] LDS
    static __device__ __forceinline__ void dma4_lane0(const void *src, float *lds_dst) {      // 4 bytes, lane 0 only
        if ((threadIdx.x & 63) == 0) {
            unsigned keep;
            const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds_dst;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
        }
    }
    // before the first DMA piece of the pipeline fill (MF only; the other flavours load inside fill())
    __device__ __forceinline__ void preload() {
        if constexpr (MF) {
            load_inputs();
            aborted = 0;
            const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
            if (chk_ctl) {                           // wave-uniform: Check(s-1), see ActArgs::chk_ctl
                for (int p = w; p * 256 < chk_n && p * 256 < CHK_MAX; p += nthreads / 64)
                    dma16(reinterpret_cast<const char *>(chk_slots + 256 * p) + 16 * lane, s_slots + 256 * p);
                if (w == 0) dma4_lane0(&chk_ctl->done, s_slots + CHK_MAX);
            } else if (skip) {
                if (w == 0) dma4_lane0(skip, s_slots + CHK_MAX);
            }
        }
    }
    __device__ __forceinline__ void fill() {
        if (!MF) load_inputs();
        if (with_rng) rng.fill();       // wave-uniform
    }
    // behind the counted wait + barrier that end the pipeline fill: everything preload() requested has landed in LDS
    __device__ __forceinline__ void post_fill() {
        if constexpr (MF) {
            if (chk_ctl) {
                // (16-byte LDS reads, all of a thread's in flight at once)
                constexpr int NR = CHK_MAX / (4 * NTH);
                f32x4 v[NR];
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    const int e = 4 * ((int)threadIdx.x + q * NTH);
                    v[q] = *reinterpret_cast<const f32x4 *>(s_slots + (e < chk_n ? e : 0));
                }
                float m = 0.f;
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    const bool in = 4 * ((int)threadIdx.x + q * NTH) < chk_n;
                    m = fmaxf(m, in ? fmaxf(fmaxf(v[q][0], v[q][1]), fmaxf(v[q][2], v[q][3])) : 0.f);
                }
                m = wave_max_lane63(m);
                if ((threadIdx.x & 63) == 63) s_chk[threadIdx.x >> 6] = m;
                wg_barrier();
                m = 0.f;
#pragma unroll
                for (int q = 0; q < NTH / 64; ++q) m = fmaxf(m, s_chk[q]);
                const int was = __float_as_int(s_slots[CHK_MAX]);
                const int done = was || !(m > chk_tol);
                if (blockIdx.x == 0 && threadIdx.x == 0 && !was) { chk_ctl->steps += 1; chk_ctl->done = done; }
                aborted = __builtin_amdgcn_readfirstlane(done);
            } else if (skip) {
                aborted = __builtin_amdgcn_readfirstlane(__float_as_int(s_slots[CHK_MAX]) != 0);
            }
        }
    }
    __device__ __forceinline__ void drain() {}
};

// act_kernel's epilogue for the lane's outputs of ONE output tile (i0, j0): activation, draw, stores, the per-row
// partial sums.  Returns the lane's mean-field residual max|m - prev| (0 without a.prev).  A function so that the
// persistent fast-binary kernel (act_bf3_kernel) can call it once per tile of its strip.
template <class G, int ABL, class SideT, bool HWMATH = false, bool FE = false, bool LIT = false>
__device__ __forceinline__ float act_epilogue(const ActArgs &a, const PhiloxKey &key, const f32x4 (&acc)[G::MI][1], const SideT &side,
                                              int i0, int j0) {
    constexpr int E = G::E, NH = G::MI;            // NH = Philox blocks (groups of 4 outputs) per lane
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w % G::WI, wj = w / G::WI;
    const int g = lane >> 4, l15 = lane & 15;
    const int ib0 = i0 + wi * (16 * G::MI) + g * E;
    const int j = j0 + wj * 16 + l15;
    const bool rng_fast = ((a.I & 3) == 0);
    const float (&bs)[E] = side.bs;
    const float (&sg)[E] = side.sg;
    const typename PhiloxFor<G::MI>::type &rng = side.rng;

    float z[E];
    lane_outputs<G>(acc, 0, z);
    float dmax = 0.f;
    if (BM_ABL(4)) {
        float zs = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) zs += z[e];
        if (j < a.J && ib0 < a.I && a.states) a.states[(size_t)j * a.ldo + ib0] = zs;
    } else if (j < a.J && ib0 < a.I) {
        const bool al_out = ((a.ldo & 3) == 0);
#pragma unroll
        for (int hlf = 0; hlf < NH; ++hlf) {
            const int ib = ib0 + 4 * hlf;
            if (ib >= a.I) break;
            const int nvalid = (a.I - ib < 4) ? a.I - ib : 4;
            float m[4], s[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = a.mult * z[4 * hlf + r];
                const float b = a.bmult * bs[4 * hlf + r];
                m[r] = (a.kind == 0) ? (LIT ? sigmoid_literal(x + b) : (HWMATH ? sigmoid_hw(x + b) : sigmoid(x + b)))
                                     : (a.kind == 1 ? (x * sg[4 * hlf + r] + b) : (a.kind == 3 ? x + b : x));
                s[r] = m[r];
            }
            if (a.sample) {
                if (rng_fast) {          // the 4 outputs are exactly one (precomputed) Philox block
                    const uint32_t *wds = rng.words(hlf);
                    if (a.kind == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[r] = (u32_to_uniform(wds[r]) < m[r]) ? 1.f : 0.f;
                    } else {
                        float n[4];
                        box_muller(wds[0], wds[1], n[0], n[1]);
                        box_muller(wds[2], wds[3], n[2], n[3]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[r] = n[r] * sg[4 * hlf + r] + m[r];
                    }
                } else {
                    const unsigned long long flat = (unsigned long long)(a.row0 + j) * (unsigned long long)a.I + ib;
                    draw4(a, key, flat, ib, nvalid, m, s, false);
                }
            }
            const size_t o = (size_t)j * a.ldo + ib;
            if constexpr (SideT::kMF) {
                if (a.prev) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r < nvalid) dmax = fmaxf(dmax, fabsf(m[r] - side.pv[4 * hlf + r]));
                }
            }
            const bool v4 = al_out && nvalid == 4;
            if (a.means) store4<HWMATH>(a.means, o, m, nvalid, v4 && (((uintptr_t)a.means & 15u) == 0));
            if (a.negmeans) {
                float nm[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) nm[r] = -m[r];
                store4<HWMATH>(a.negmeans, o, nm, nvalid, v4 && (((uintptr_t)a.negmeans & 15u) == 0));
            }
            if (a.states) store4<HWMATH>(a.states, o, s, nvalid, v4 && (((uintptr_t)a.states & 15u) == 0));
            if (HWMATH && a.states16) {        // fast-binary strip kernel only: bf16 shadow of the {0,1} states (exact),
                                               // pitch ld16 % 64 == 0; fp32 launches get it from shadow16_kernel (launch_act)
                uint16_t *d = a.states16 + (size_t)j * a.ld16 + ib;
                if (nvalid == 4 && (ib & 3) == 0) {
                    uint2 pk;
                    pk.x = (__float_as_uint(s[0]) >> 16) | (__float_as_uint(s[1]) & 0xffff0000u);
                    pk.y = (__float_as_uint(s[2]) >> 16) | (__float_as_uint(s[3]) & 0xffff0000u);
                    *reinterpret_cast<uint2 *>(d) = pk;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (r < nvalid) d[r] = (uint16_t)(__float_as_uint(s[r]) >> 16);
                }
            }
        }
    }
    // (FE: the h0 pass of a fused metric fetch - a compile-time flavour of its own (act_kernel<..., FE = true>): as a runtime
    //  branch of the shared epilogue the extra code cost every propagation pass 0.1 - 0.2 us, 0.5 us per CD-1 update, same-box A/B)
    if constexpr (FE) { if (a.fe_zero && i0 == 0 && j0 == 0 && tid < 6) a.fe_zero[tid] = 0.0; }     // the first tile zeroes the fetch's accumulators
    if (a.rowacc || a.rowdot_out) {           // wave-uniform
        // per-lane quads (4 consecutive i, left to right); MI == 2: the lane's two quads are added
        float racc = 0.f, rdot = 0.f, racc2 = 0.f;
        if (j < a.J) {
            int fc = 0; float delta = 0.f;
            if constexpr (FE) {
                fc = (a.fe_flip == FE_FLIP_FROM_KEY) ? pll_flip_col(a.fe_key, (unsigned long long)(a.row0 + j), a.K1) : a.fe_flip[j];
                delta = 1.0f - 2.0f * a.fe_x[(size_t)j * a.fe_ldx + fc];
            }
#pragma unroll
            for (int hlf = 0; hlf < NH; ++hlf) {
                float qa = 0.f, qd = 0.f, qa2 = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = 4 * hlf + r, i = ib0 + e;
                    if (i >= a.I) break;
                    if (a.rowacc) {
                        if (a.dot_mat) {
                            qa += z[e] * a.dot_mat[(size_t)j * a.ld_dot + i];
                        } else {
                            const float t = z[e] + bs[e];
                            if constexpr (FE) qa2 += softplus(t + delta * a.fe_w[(size_t)fc * a.fe_ldw + i]);    // the PLL partner's row (beta_b == 1)
                            if (a.rowacc_single) qa += HWMATH ? softplus_hw(a.beta_b * t) : softplus(a.beta_b * t);
                            else qa += softplus_hw(a.beta_b * t) - softplus_hw(a.beta_a * t);
                        }
                    }
                    // the state of this element as it was just stored by this lane
                    if (a.rowdot_out) qd += a.states[(size_t)j * a.ldo + i] * a.dot_vec[i];
                }
                racc = (hlf == 0) ? qa : racc + qa;
                rdot = (hlf == 0) ? qd : rdot + qd;
                racc2 = (hlf == 0) ? qa2 : racc2 + qa2;
            }
        }
        // 16-column slot sums: MI == 1: lanes g = 0..3 hold q0..q3 -> (q0+q1)+(q2+q3);
        // MI == 2: lanes g hold (q_2g + q_2g+1); g in {0,1} / {2,3} are the wave's two slots
        racc += __shfl_xor(racc, 16);
        rdot += __shfl_xor(rdot, 16);
        if (G::MI == 1) { racc += __shfl_xor(racc, 32); rdot += __shfl_xor(rdot, 32); }
        if constexpr (FE) { racc2 += __shfl_xor(racc2, 16); if (G::MI == 1) racc2 += __shfl_xor(racc2, 32); }
        const bool writer = (G::MI == 1) ? (g == 0) : ((g & 1) == 0);
        const int slot = (i0 + wi * (16 * G::MI)) / 16 + ((G::MI == 2) ? (g >> 1) : 0);
        if (writer && j < a.J && slot * 16 < a.I) {
            if constexpr (FE) {          // metric fetch: row-major partials
                a.rowacc[(size_t)j * a.fe_rm + slot] = racc;
                a.fe_rowacc2[(size_t)j * a.fe_rm + slot] = racc2;
            } else {
                if (a.rowacc) a.rowacc[(size_t)slot * a.ld_part + j] = racc;
                if (a.rowdot_out) a.rowdot_out[(size_t)slot * a.ld_part + j] = rdot;
            }
        }
    }
    return dmax;
}

// MINB: HIP's second __launch_bounds__ argument = WAVES PER SIMD the register budget must allow (for the 4-wave
// geometries that equals the workgroups per CU; an 8-wave workgroup that should run twice per CU passes 4)
template <class G, int MINB, bool SEG2, bool FAST, int ABL = 0, int PL = KM, int STG = STG_DMA, bool FE = false, bool LIT = false, bool MF = false>
__global__ __launch_bounds__(G::NT, MINB) void act_kernel(ActArgs a, TileMap tmap) {
    // (the block -> tile map is an argument of its own: the grid path indexes it with blockIdx & 7, and a dynamically
    //  indexed member made hipcc fetch EVERY ActArgs field lazily in small pieces - 50 scalar loads with their waits
    //  instead of 22, +1 us on the prop-down; round 4, same-box A/B)
    __shared__ __attribute__((aligned(16))) float smem[G::SMEM_FLOATS];
    constexpr int E = G::E;
    BM_STAMP(0);
    // All hot kernel arguments in SGPRs after ONE scalar-memory round trip (hipcc otherwise
    // loads them lazily: five serialized kernarg waits before the first operand load goes out).
    asm volatile("" :: "s"(a.P1.ptr), "s"(a.Q1.ptr), "s"(a.P1.ld), "s"(a.Q1.ld), "s"(a.P1.nx), "s"(a.Q1.nx),
                       "s"(a.K1), "s"(a.K2), "s"(a.bias), "s"(a.sigma), "s"(a.means), "s"(a.states), "s"(a.ldo),
                       "s"(a.sample), "s"(a.kind), "s"(a.row0), "s"(a.I), "s"(a.J), "s"(a.skip));
    const int tiles_j = (a.J + G::TJ - 1) / G::TJ;
    int ti, tj;
    // (the mean-field plumbing - ActArgs::chk_ctl / skip / prev / maxdiff / acc_init - exists in the MF flavour only:
    //  launch_act routes every launch that carries one of them there, ActSide<.., MF> says how it is scheduled)
    // tile order: 2-D XCD rectangles, L2-sized column groups (tile_of_block)
    if (tmap.slab) {                               // wave-uniform; the launch tuner's choice per shape (TileMap::slab)
        block_to_tile(tiles_j, ti, tj, 0, 0, a.J > 2 * a.I, (a.I + G::TI - 1) / G::TI);
    } else {
        TILE_WORDS(tmap, (int)blockIdx.x);
        TILE_OF_BLOCK(tmap, (int)blockIdx.x, (int)gridDim.x, ti, tj);
    }
    const int i0 = ti * G::TI, j0 = tj * G::TJ;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w % G::WI, wj = w / G::WI;
    const int g = lane >> 4, l15 = lane & 15;
    const int ib0 = i0 + wi * (16 * G::MI) + g * E;     // E consecutive outputs i = ib0 + e

    KRange kr;
    kr.P1 = a.P1; kr.Q1 = a.Q1; kr.K1 = a.K1;
    kr.P2 = a.P2; kr.Q2 = a.Q2; kr.K2 = a.K2;
    static_assert(G::NJ == 1, "act_kernel: one j sub-tile per wave");
    const int j = j0 + wj * 16 + l15;
    // Side work for the pipeline fill (runs while the first operand loads are in flight):
    // the epilogue inputs (bias, sigma) and, when a draw follows, the lane's Philox block(s).
    ActSide<E, typename PhiloxFor<G::MI>::type, MF, G::NT> side;
    side.bias = a.bias; side.sigma = a.sigma; side.ib0 = ib0; side.I = a.I; side.with_rng = a.sample;
    side.prev_row = (MF && a.prev && j < a.J && ib0 < a.I) ? a.prev + (size_t)j * a.ldo + ib0 : nullptr;
    const PhiloxKey key = a.key;
    side.rng.init(key, ((unsigned long long)(a.row0 + j) * (unsigned long long)a.I + ib0) >> 2);

    f32x4 acc[G::MI][1];
#pragma unroll
    for (int t = 0; t < G::MI; ++t) acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (MF) {
        __shared__ float s_chk[G::NT / 64];
        __shared__ __attribute__((aligned(16))) float s_slots[ActSide<E, typename PhiloxFor<G::MI>::type, true>::CHK_MAX + 4];
        side.s_slots = s_slots;
        side.chk_ctl = a.chk_ctl; side.chk_slots = a.chk_slots; side.chk_n = a.chk_n; side.chk_tol = a.chk_tol;
        side.skip = a.skip; side.s_chk = s_chk; side.nthreads = G::NT; side.aborted = 0;
        if (a.acc_init && j < a.J) {               // start the chain from a stored partial sum
            const float *src = a.acc_init + (size_t)j * a.ld_init + ib0;
            if (G::MI == 1 && ib0 + 3 < a.I && (((uintptr_t)src & 15u) == 0)) {
                const float4 t4 = *reinterpret_cast<const float4 *>(src);
                acc[0][0] = (f32x4){t4.x, t4.y, t4.z, t4.w};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < G::MI; ++t) {
                        const int i = ib0 + G::MI * r + t;
                        if (i < a.I) acc[t][0][r] = src[G::MI * r + t];
                    }
            }
        }
        side.preload();
    }
#ifdef BM_PROBE
    mainloop<XM, G, FAST, SEG2, ABL, PL, STG>(acc, kr, i0, j0, smem, side, a.dbg ? a.dbg + 2048 + blockIdx.x * 8 : nullptr);
#else
    mainloop<XM, G, FAST, SEG2, ABL, PL, STG>(acc, kr, i0, j0, smem, side);
#endif
    BM_STAMP(1);
    if constexpr (MF) { if (side.aborted) return; }          // the loop had ended: nothing is written
    float dmax = act_epilogue<G, ABL, decltype(side), false, FE, LIT>(a, key, acc, side, i0, j0);
    if (MF && a.maxdiff) {     // wave-uniform.  ONE atomic per workgroup: thousands of same-address atomics
                               // (one per wave) serialise in the L2 and doubled the duration of the sweep kernels
        __shared__ float s_wavemax[G::NT / 64];
        dmax = wave_max_lane63(dmax);
        if (lane == 63) s_wavemax[w] = dmax;
        wg_barrier();               // LDS-only hand-over: no fence that waits for the global stores / loads in flight
        if (tid == 0) {
            float m = 0.f;
#pragma unroll
            for (int q = 0; q < G::NT / 64; ++q) m = fmaxf(m, s_wavemax[q]);
            if (a.maxdiff_blk && gridDim.x <= BM_MF_SLOTS) a.maxdiff_blk[blockIdx.x] = m;
            else if (m > 0.f) atomicMax(a.maxdiff, __float_as_uint(m));
        }
    }
    BM_STAMP(2);
}

// ----------------------------------------------------------- act_bf3_kernel (fast-binary mode, bm_bf3.h)
// The same propagation + activation + draw from bf16 weight planes and bf16 state shadows, as a PERSISTENT strip
// kernel: the bf16 matrix cores run a 64-k chunk in ~0.1 us, far below a memory round trip, so a tile-per-workgroup
// launch spends its time in the pipeline fill and the epilogue of every tile (measured: 11.7 us per 64 x 64 tile of
// the AIS visible update, 1.3 us of it matrix time).  Here a workgroup owns tile row ti and a strip of tile columns
// [tj0, tj1): the first chunks of tile tj+1 are requested BEFORE the epilogue of tile tj runs, the bias / sigma
// loads and the DMA plan of the weight planes are done once per strip.
struct Bf3Strip { int tiles_i, tiles_j, strips, abl; };  // grid = tiles_i * strips; strip s of row ti: blocks ti * strips + s
                                                         // abl (BM355_DEBUG=bf3_abl, measurements only): 1 no epilogue, 2 no K loop

template <class G, bool SEG2, int MINW>
__global__ __launch_bounds__(G::NT, MINW) void act_bf3_kernel(ActArgs a, Bf3Strip sp) {
    __shared__ __attribute__((aligned(16))) float smem[Bf3Geo<G>::SMEM_FLOATS];
    constexpr int E = G::E;
    const int ti = (int)blockIdx.x / sp.strips, st = (int)blockIdx.x % sp.strips;
    const int per = sp.tiles_j / sp.strips, rem = sp.tiles_j % sp.strips;
    const int tj0 = st * per + (st < rem ? st : rem), tj1 = tj0 + per + (st < rem ? 1 : 0);
    const int i0 = ti * G::TI;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w % G::WI, wj = w / G::WI;
    const int g = lane >> 4, l15 = lane & 15;
    const int ib0 = i0 + wi * (16 * G::MI) + g * E;
    ActSide<E, typename PhiloxFor<G::MI>::type> side;
    side.bias = a.bias; side.sigma = a.sigma; side.ib0 = ib0; side.I = a.I; side.with_rng = 0;
    side.prev_row = nullptr;
    side.fill();                                   // bias / sigma of the lane's outputs: once per strip
    side.with_rng = a.sample;
    Bf3Pipe<G, SEG2> pipe;
    pipe.setup(a.b3, i0, smem);
    if (tj0 < tj1) { pipe.set_tile(a.b3, tj0 * G::TJ); pipe.prefetch(); }
    for (int tj = tj0; tj < tj1; ++tj) {
        const int j0 = tj * G::TJ, j = j0 + wj * 16 + l15;
        f32x4 acc[G::MI][1];
#pragma unroll
        for (int t = 0; t < G::MI; ++t) acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        side.rng.init(a.key, ((unsigned long long)(a.row0 + j) * (unsigned long long)a.I + ib0) >> 2);
        if (a.sample) side.rng.fill();             // the lane's Philox blocks, while the tile's first chunks arrive
        if (!(sp.abl & 2)) pipe.run(acc);
        if (tj + 1 < tj1) { pipe.set_tile(a.b3, (tj + 1) * G::TJ); pipe.prefetch(); }     // next tile's pipeline fill ...
        if (!(sp.abl & 1)) (void)act_epilogue<G, 0, decltype(side), true>(a, a.key, acc, side, i0, j0);   // ... under this tile's epilogue
    }
}

// -------------------------------------------------------------- colstat_kernel
// Column sums in canonical order via MFMA against ones:
//   D[i][*] = sum_k A[k][i] * 1  ==  sequential fp32 sum over rows k.
// One wave per 16 columns, operands straight from global memory (64 B per row).
// job: out[c] = sum_b (A[b][c] - Bm[b][c])   (Bm may be null -> plain column sum)
struct ColSumJob {
    const float *A, *Bm;
    int lda, ldb, ncols, nrows;
    float *out;
};
constexpr int MAX_COLJOBS = 12;
struct ColSumArgs {
    ColSumJob job[MAX_COLJOBS];
    int first_wave[MAX_COLJOBS + 1];   // prefix sums of ceil(ncols/16)
    int njobs;
};

// Canonical column sums of 64 columns [c0, c0+64) by one 256-thread workgroup; wave w owns the
// 16 columns c0 + 16w ...:
//   sum1[c] = sum_k (A - Bm)[k][c]          sum2[c] = sum_k Bm[k][c]   (optional)
// The MFMA chain D[i][*] += A[k][i] * 1 is sequential in k (that IS the canonical order), so the
// parallelism is across columns and in the loads: all 4 waves stage CS_ROWS rows x 64 columns of
// both operands into LDS with 16-byte loads (the next chunk is already in flight in registers
// while the current one is consumed), then every wave runs its chain(s) out of LDS (row stride
// 80 floats: lanes 0-15 / 16-31 hit disjoint bank halves).  Results land in the lanes with
// (lane & 15) == 0: acc[r] is column c0 + 16w + 4*(lane>>4) + r.
constexpr int CS_ROWS = 128;
constexpr int CS_LD = 80;
constexpr int CS_SMEM_FLOATS = 2 * CS_ROWS * CS_LD;    // 80 KiB

constexpr int CS_NV = CS_ROWS * 16 / NT;     // float4 per thread per operand (8)
struct ColRegs { float4 a[CS_NV], b[CS_NV]; };    // (a struct of fixed arrays + unrolled loops stays in VGPRs;
                                                  //  lambdas capturing the arrays by reference went to scratch)

// rows [r0, r0 + CS_ROWS) x 64 columns of both operands -> registers (clamped, branch-free)
__device__ __forceinline__ void cs_fetch(ColRegs &r, const float *A, int lda, const float *Bm, int ldb,
                                         int c0, int ncols, int nrows, int r0, int tid) {
    const int c4 = tid & 15;
    const int cc = min(c0 + 4 * c4, ncols - 4);
#pragma unroll
    for (int n = 0; n < CS_NV; ++n) {
        const int rc = min(r0 + (tid >> 4) + 16 * n, nrows - 1);
        r.a[n] = *reinterpret_cast<const float4 *>(A + (size_t)rc * lda + cc);
        r.b[n] = Bm ? *reinterpret_cast<const float4 *>(Bm + (size_t)rc * ldb + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// registers -> LDS, rows past nrows / columns past ncols zeroed
__device__ __forceinline__ void cs_stash(const ColRegs &r, float *sA, float *sB, int c0, int ncols, int nrows,
                                         int r0, int tid) {
    const int c4 = tid & 15;
#pragma unroll
    for (int n = 0; n < CS_NV; ++n) {
        const int row = (tid >> 4) + 16 * n;
        const bool ok = (r0 + row < nrows) && (c0 + 4 * c4 < ncols);
        // (select VALUES: `ok ? r.a[n] : z` on lvalues selects a pointer and forces the set into scratch)
        float4 va = r.a[n], vb = r.b[n];
        if (!ok) { va = make_float4(0.f, 0.f, 0.f, 0.f); vb = va; }
        *reinterpret_cast<float4 *>(sA + row * CS_LD + 4 * c4) = va;
        *reinterpret_cast<float4 *>(sB + row * CS_LD + 4 * c4) = vb;
    }
}

// negB: Bm holds the NEGATED second operand (act_kernel's `negmeans`): a + (-b) == a - b and
// -(sum of -b) == sum of b exactly, so the results are bit-identical to the plain form.
__device__ __forceinline__ void block_colsum(const float *A, int lda, const float *Bm, int ldb,
                                             int c0, int ncols, int nrows, bool want2,
                                             float *smem, f32x4 &sum1, f32x4 &sum2, bool negB = false) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    float *sA = smem, *sB = smem + CS_ROWS * CS_LD;
    sum1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    sum2 = (f32x4){0.f, 0.f, 0.f, 0.f};
    // 16-byte path: aligned operands and whole float4s (a ragged last column group is handled by
    // clamping the column and zeroing at the LDS store, not by the scalar path)
    const bool vec = (((uintptr_t)A & 15u) == 0) && ((lda & 3) == 0) && ((ncols & 3) == 0) &&
                     (!Bm || ((((uintptr_t)Bm & 15u) == 0) && ((ldb & 3) == 0)));
    ColRegs regs;
    if (vec) cs_fetch(regs, A, lda, Bm, ldb, c0, ncols, nrows, 0, tid);
    for (int r0 = 0; r0 < nrows; r0 += CS_ROWS) {
        const int nr = (nrows - r0 < CS_ROWS) ? nrows - r0 : CS_ROWS;
        if (vec) {
            cs_stash(regs, sA, sB, c0, ncols, nrows, r0, tid);
        } else {
            for (int e = tid; e < CS_ROWS * 64; e += NT) {
                const int row = e >> 6, cc = e & 63, c = c0 + cc;
                const bool ok = row < nr && c < ncols;
                sA[row * CS_LD + cc] = ok ? A[(size_t)(r0 + row) * lda + c] : 0.f;
                sB[row * CS_LD + cc] = (ok && Bm) ? Bm[(size_t)(r0 + row) * ldb + c] : 0.f;
            }
        }
        wg_barrier();               // LDS-only hand-over: no fence that waits for the global stores / loads in flight
        // next chunk in flight under the chain (clamped loads are legal for any r0)
        if (vec) cs_fetch(regs, A, lda, Bm, ldb, c0, ncols, nrows, r0 + CS_ROWS, tid);
        // rows >= nr are zero in LDS, so the chain may run to a multiple of 8 steps:
        // 8 fragment reads are issued ahead of the 8 dependent MFMAs that consume them
        const int nsteps = (((nr + 3) / 4) + 7) & ~7;
        const int co = w * 16 + l15;
        for (int s = 0; s < nsteps; s += 8) {
            float d[8], e[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int o = (4 * (s + u) + g) * CS_LD + co;
                e[u] = sB[o];
                d[u] = negB ? sA[o] + e[u] : sA[o] - e[u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                sum1 = __builtin_amdgcn_mfma_f32_16x16x4f32(d[u], 1.0f, sum1, 0, 0, 0);
                if (want2) sum2 = __builtin_amdgcn_mfma_f32_16x16x4f32(e[u], 1.0f, sum2, 0, 0, 0);
            }
        }
        wg_barrier();               // LDS-only hand-over: no fence that waits for the global stores / loads in flight
    }
    if (negB) sum2 = -sum2;
}

__global__ __launch_bounds__(NT) void colsum_kernel(ColSumArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[CS_SMEM_FLOATS];
    const int wv = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int jb = 0;
    while (jb + 1 < a.njobs && wv >= a.first_wave[jb + 1]) ++jb;
    const ColSumJob J = a.job[jb];
    const int c0 = (wv - a.first_wave[jb]) * 64;
    f32x4 s1, s2;
    block_colsum(J.A, J.lda, J.Bm, J.ldb, c0, J.ncols, J.nrows, false, smem, s1, s2);
    if ((lane & 15) == 0) {
        const int g = lane >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cc = c0 + w * 16 + g * 4 + r;
            if (cc < J.ncols) J.out[cc] = s1[r];
        }
    }
}

// RBM bias / running-mean update from raw column sums (base_rbm.py:450-474).
//   sv[c] = sum_b (X - v_k),  sh[c] = sum_b (h0 - h_k),  sq[c] = sum_b h_k
struct RbmBiasArgs {
    const float *sv, *sh, *sq;
    float *vb, *dvb, *hb, *dhb, *q, *pen;
    int V, H;
    float N, lr, mom, damping, cost, target;
};
__global__ void rbm_bias_kernel(RbmBiasArgs a) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < a.V) {
        const float g = a.sv[c] / a.N;                       // reduce_mean(X - v, 0)     :451
        const float d = a.lr * (a.mom * a.dvb[c] + g);       // :470
        a.dvb[c] = d;
        a.vb[c] = a.vb[c] + d;                               // :471
    } else if (c < a.V + a.H) {
        const int h = c - a.V;
        const float qn = a.damping * a.q[h] + (1.0f - a.damping) * a.sq[h];   // :457-459 (column SUM)
        a.q[h] = qn;
        const float pen = a.cost * (qn - a.target);          // :460
        a.pen[h] = pen;
        float g = a.sh[h] / a.N;                             // :453
        g = g - pen;                                         // :461
        const float d = a.lr * (a.mom * a.dhb[h] + g);       // :473
        a.dhb[h] = d;
        a.hb[h] = a.hb[h] + d;                               // :474
    }
}

// Fused single-GPU form of colsum_kernel + rbm_bias_kernel: the wave that sums a group
// of 16 columns applies their bias / q_means update directly (identical arithmetic).
struct RbmBiasFusedArgs {
    const float *X, *vs, *h0m, *hm;     // [B][V] pitch ldx / ldv, [B][H] pitch ldh0 / ldh
    int ldx, ldv, ldh0, ldh, B;
    int hm_negated;                     // 1: `hm` points at -h_k (the negmeans buffer)
    float *raw_tail;                    // [V | H | H] raw sums are still published (metrics / tests)
    int raw_only;                       // 1: publish the raw sums only (data-parallel phase 1), no update
    RbmBiasArgs u;
};
// body of one 64-column group (block index wv); smem >= CS_SMEM_FLOATS floats
__device__ __forceinline__ void rbm_bias_fused_block(const RbmBiasFusedArgs &a, int wv, float *smem) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4;
    const int nv = (a.u.V + 63) / 64;
    f32x4 s1, s2;
    if (wv < nv) {
        const int c0 = wv * 64;
        block_colsum(a.X, a.ldx, a.vs, a.ldv, c0, a.u.V, a.B, false, smem, s1, s2);   // sum(X - v_k)
        if ((lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = c0 + w * 16 + g * 4 + r;
                if (c < a.u.V) {
                    a.raw_tail[c] = s1[r];
                    if (a.raw_only) continue;
                    const float gr = s1[r] / a.u.N;
                    const float d = a.u.lr * (a.u.mom * a.u.dvb[c] + gr);
                    a.u.dvb[c] = d;
                    a.u.vb[c] = a.u.vb[c] + d;
                }
            }
        }
    } else {
        const int c0 = (wv - nv) * 64;
        block_colsum(a.h0m, a.ldh0, a.hm, a.ldh, c0, a.u.H, a.B, true, smem, s1, s2, a.hm_negated != 0);  // sum(h0 - h_k), sum(h_k)
        if ((lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int h = c0 + w * 16 + g * 4 + r;
                if (h < a.u.H) {
                    const float sq = s2[r];
                    a.raw_tail[a.u.V + h] = s1[r];
                    a.raw_tail[a.u.V + a.u.H + h] = sq;
                    if (a.raw_only) continue;
                    const float qn = a.u.damping * a.u.q[h] + (1.0f - a.u.damping) * sq;
                    a.u.q[h] = qn;
                    const float pen = a.u.cost * (qn - a.u.target);
                    a.u.pen[h] = pen;
                    float gr = s1[r] / a.u.N;
                    gr = gr - pen;
                    const float d = a.u.lr * (a.u.mom * a.u.dhb[h] + gr);
                    a.u.dhb[h] = d;
                    a.u.hb[h] = a.u.hb[h] + d;
                }
            }
        }
    }
}

__global__ __launch_bounds__(NT) void rbm_bias_fused_kernel(RbmBiasFusedArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[CS_SMEM_FLOATS];
    rbm_bias_fused_block(a, blockIdx.x, smem);
}

// ----------------------------------------------------------------- grad_kernel
// out[j][i] over i = above-units (contiguous in W), j = below-units; K = rows.
struct GradArgs {
    Operand Ppos, Qpos; int Kpos;     // positive phase:  sum_b Qpos[b][j] * Ppos[b][i]
    Operand Pneg, Qneg; int Kneg;     // negative phase (form 0: Pneg holds the NEGATED means, see below)
    int I, J;
    int form;                         // 0: (pos-neg)/N  (RBM, base_rbm.py:447-449); 1: pos/N - neg/M (DBM, dbm.py:553-570)
    int fused;                        // 1: apply the update in the epilogue; 0: write raw sums to `raw`
    float *raw;                       // [J][I] raw pos-neg (form 0) ; for form 1: raw pos at raw, raw neg at raw2
    float *raw2;
    float *W, *dW;                    // [J][I], pitch ldw (raw/raw2 share it)
    float *Wt;                        // [I][J] transpose of W kept in sync (or null), pitch ldwt
    int ldw, ldwt;
    const float *pen;                 // [I] sparsity penalty (already cost*(q-target) [+ mu term]) or null
    float N, M, l2, lr, mom;
    // RBM single-GPU fusion: the LAST `nbias` workgroups of the launch run the column-sum
    // + bias update (rbm_bias_fused_block) concurrently with the tile workgroups.  Legal
    // only when the W update does not need the penalty they produce (sparsity_cost == 0).
    int nbias;
    RbmBiasFusedArgs bias;
    int fetch_at_fill;                // 1: read W/dW of the lane's outputs during the pipeline fill (set by launch_grad)
    int map_xi;                       // see ActArgs::map_xi
#ifdef BM_PROBE
    long long *dbg;
#endif
};

// the scalar update rule shared by the fused epilogue and the split (data-parallel) apply kernel
// (explicit round-to-nearest intrinsics: the replicas of a data-parallel run and the one-engine run apply this formula in
//  DIFFERENT kernels - apply_w_kernel, apply_w_tiled_kernel, grad_kernel's epilogue, bm_xchg.hip's fused exchanges - and must
//  produce the same bits whatever a toolchain's -ffp-contract default is; round-5 advisor)
__device__ __forceinline__ void apply_w_update(float g, float pen, float l2, float lr, float mom,
                                               float &w, float &dw) {
    g = __fsub_rn(g, __fmul_rn(l2, w));                 // dW = ... - l2*W            (base_rbm.py:449)
    g = __fsub_rn(g, pen);                              // dW -= sparsity_penalty      (base_rbm.py:462)
    const float d = __fmul_rn(lr, __fadd_rn(__fmul_rn(mom, dw), g));   // dW_update  (base_rbm.py:467)
    dw = d;
    w = __fadd_rn(w, d);                                // W.assign_add               (base_rbm.py:468)
}
// the gradient such an update starts from: raw / N (RBM: the outer products already hold pos - neg) or raw / N - raw2 / M
// (DBM: positive phase over N rows, negative over M particles, dbm.py:553-570).  A power-of-two count turns the division
// into an exact scaling by its reciprocal (same bits as the IEEE division).  ONE definition for every kernel above.
__device__ __forceinline__ bool grad_pow2(float N) { return ((__float_as_uint(N) & 0x007fffffu) == 0u) && N >= 1.0f; }
__device__ __forceinline__ float grad_norm1(float r, float N, float invN, bool pow2) {
    return pow2 ? __fmul_rn(r, invN) : __fdiv_rn(r, N);
}
__device__ __forceinline__ float grad_norm2(float r, float r2, float N, float M, float invN, float invM, bool pow2) {
    return pow2 ? __fsub_rn(__fmul_rn(r, invN), __fmul_rn(r2, invM)) : __fsub_rn(__fdiv_rn(r, N), __fdiv_rn(r2, M));
}

// x / N, bit-identical to the IEEE division: a power-of-two N (the usual batch sizes) turns it
// into an exact scaling by 1/N; anything else takes the real division.
struct DivBy {
    float n, inv; bool pow2;
    __device__ __forceinline__ explicit DivBy(float N) : n(N), inv(1.0f / N),
        pow2(grad_pow2(N)) {}
};

// grad_kernel's side work: the W / dW values of the lane's outputs are fetched at the start of
// the pipeline drain, so the read-modify-write epilogue does not start with a memory round trip
template <int NJ> struct GradSide {
    static constexpr bool kFinalSync = true;     // form 1 runs two pipelines through the same LDS ring
    static constexpr bool kSplitFill = false, kCohQ = false, kCanAbort = false;
    const float *W, *dW; int ldw, I, J, ib0, jb[NJ]; bool on, vec8;
    float4 w[NJ][2], d[NJ][2];
    bool at_fill;
    __device__ __forceinline__ void fill() { if (at_fill) fetch(); }
    __device__ __forceinline__ void drain() { if (!at_fill) fetch(); }
    __device__ __forceinline__ void fetch() {
        if (!on || ib0 >= I) return;
#pragma unroll
        for (int n = 0; n < NJ; ++n) {
            if (jb[n] >= J) continue;
            const size_t o = (size_t)jb[n] * ldw + ib0;
            if (vec8) {
                w[n][0] = *reinterpret_cast<const float4 *>(W + o);  w[n][1] = *reinterpret_cast<const float4 *>(W + o + 4);
                d[n][0] = *reinterpret_cast<const float4 *>(dW + o); d[n][1] = *reinterpret_cast<const float4 *>(dW + o + 4);
            } else {
                float tw[8], td[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool ok = ib0 + e < I;
                    tw[e] = ok ? W[o + e] : 0.f;
                    td[e] = ok ? dW[o + e] : 0.f;
                }
                w[n][0] = make_float4(tw[0], tw[1], tw[2], tw[3]); w[n][1] = make_float4(tw[4], tw[5], tw[6], tw[7]);
                d[n][0] = make_float4(td[0], td[1], td[2], td[3]); d[n][1] = make_float4(td[4], td[5], td[6], td[7]);
            }
        }
    }
};

template <class G, bool FAST, int ABL = 0, int STG = STG_DMA, int MINB = 1>
__global__ __launch_bounds__(G::NT, MINB) void grad_kernel(GradArgs a, TileMap tmap) {
    constexpr int TI = G::TI, NJ = G::NJ;
    static_assert(G::MI == 2 && G::TI == 64 && G::TJ == 64, "grad_kernel: 64 x 64 tiles, 8 consecutive outputs per lane");
    __shared__ __attribute__((aligned(16))) float smem[G::SMEM_FLOATS];
    constexpr bool kBias = G::SMEM_FLOATS >= CS_SMEM_FLOATS;      // the bias path reuses the tile LDS (launchers: nbias == 0 otherwise)
    static_assert(G::NT >= NT, "bias path");
    // the bias/colsum workgroups sit BEHIND the tile workgroups in dispatch order: 208 tiles
    // (784x1024) take 208 CUs for the whole launch, the short bias groups cycle through
    // the CUs that are left and finish inside the tiles' shadow.
    const int ntile_blocks = (int)gridDim.x - a.nbias;
    if ((int)blockIdx.x >= ntile_blocks) {
        if constexpr (kBias) {
            if (G::NT > NT && threadIdx.x >= NT) return;      // the column-sum body is written for NT threads (whole waves leave)
            rbm_bias_fused_block(a.bias, (int)blockIdx.x - ntile_blocks, smem);
        }
        return;
    }
    // hot kernel arguments after ONE scalar-memory round trip (see act_kernel)
    asm volatile("" :: "s"(a.Ppos.ptr), "s"(a.Qpos.ptr), "s"(a.Ppos.ld), "s"(a.Qpos.ld), "s"(a.Ppos.nx), "s"(a.Qpos.nx),
                       "s"(a.Pneg.ptr), "s"(a.Qneg.ptr), "s"(a.Pneg.ld), "s"(a.Qneg.ld), "s"(a.Kpos), "s"(a.Kneg),
                       "s"(a.I), "s"(a.J), "s"(a.W), "s"(a.dW), "s"(a.ldw), "s"(a.form), "s"(a.fused));
    constexpr int TJ2 = G::TJ;                    // NJ = 2: 64 x 64 tiles
    int ti, tj;
    if (tmap.slab) {                               // wave-uniform (TileMap::slab)
        block_to_tile((a.J + TJ2 - 1) / TJ2, ti, tj, 0, a.nbias);
    } else {
        TILE_WORDS(tmap, (int)blockIdx.x);
        TILE_OF_BLOCK(tmap, (int)blockIdx.x, ntile_blocks, ti, tj);
    }
    const int i0 = ti * TI, j0 = tj * TJ2;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w % G::WI, wj = w / G::WI;
    const int g = lane >> 4, l15 = lane & 15;
    const int ib0 = i0 + wi * 32 + g * 8;

#ifdef BM_PROBE
#define BM_GSTAMP(n) do { if (a.dbg && threadIdx.x == 0) a.dbg[blockIdx.x * 4 + (n)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define BM_GSTAMP(n) do {} while (0)
#endif
    BM_GSTAMP(0);
    f32x4 pos[2][NJ], neg[2][NJ];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int n = 0; n < NJ; ++n) pos[t][n] = neg[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    GradSide<NJ> side;
    side.W = a.W; side.dW = a.dW; side.ldw = a.ldw; side.I = a.I; side.J = a.J; side.ib0 = ib0;
#pragma unroll
    for (int n = 0; n < NJ; ++n) side.jb[n] = j0 + wj * (16 * NJ) + lane_j<KM, G>(l15, n);
    side.vec8 = (ib0 + 7 < a.I) && ((a.ldw & 3) == 0);            // 16-byte aligned run of 8 (ib0 % 8 == 0)
    // When are W/dW of the lane's outputs read?  In the epilogue by default; fetching them in the pipeline
    // fill (fetch_at_fill) or at the start of the drain only moved the time or lost (launch_grad).
    side.at_fill = true;
    side.on = a.fused != 0 && a.fetch_at_fill != 0;
    KRange kr;
    kr.P1 = a.Ppos; kr.Q1 = a.Qpos; kr.K1 = a.Kpos;
    if (a.form == 0) {
        // RBM: ONE chain, positive rows then negative rows with the product negated: the caller
        // passes Pneg = -h_k (act_kernel's `negmeans` output), fma(-p, q, acc) == acc - p*q exactly
        // (canonical order of the raw CD gradient, oracle: orc_rbm_raw_grads)
        kr.P2 = a.Pneg; kr.Q2 = a.Qneg; kr.K2 = a.Kneg;
        mainloop<KM, G, FAST, true, ABL, KM, STG>(pos, kr, i0, j0, smem, side);
    } else {
        // DBM: pos/N - neg/M with N != M needs the two sums separately
        kr.P2 = a.Ppos; kr.Q2 = a.Qpos; kr.K2 = 0;
        mainloop<KM, G, FAST, false, 0, KM, STG>(pos, kr, i0, j0, smem, side);
        kr.P1 = a.Pneg; kr.Q1 = a.Qneg; kr.K1 = a.Kneg;
        mainloop<KM, G, FAST, false, 0, KM, STG>(neg, kr, i0, j0, smem, side);
    }

    BM_GSTAMP(1);
    if (ib0 >= a.I) return;
    if (!a.fetch_at_fill) { side.on = a.fused != 0; side.fetch(); }
    const DivBy divN(a.N), divM(a.M);
    float wt[NJ][8];                // updated W values of the lane's j (NJ == 2: adjacent columns of Wt)
    bool jok[NJ];
#pragma unroll
    for (int n = 0; n < NJ; ++n) {
        const int j = side.jb[n];
        jok[n] = j < a.J;
        if (!jok[n]) continue;
        float pv[8], nv[8];
        lane_outputs<G>(pos, n, pv);
        lane_outputs<G>(neg, n, nv);
        const size_t o = (size_t)j * a.ldw + ib0;
        if (!a.fused) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (ib0 + e >= a.I) break;
                a.raw[o + e] = pv[e];
                if (a.form != 0) a.raw2[o + e] = nv[e];
            }
            continue;
        }
        float wv[8] = {side.w[n][0].x, side.w[n][0].y, side.w[n][0].z, side.w[n][0].w, side.w[n][1].x, side.w[n][1].y, side.w[n][1].z, side.w[n][1].w};
        float dv[8] = {side.d[n][0].x, side.d[n][0].y, side.d[n][0].z, side.d[n][0].w, side.d[n][1].x, side.d[n][1].y, side.d[n][1].z, side.d[n][1].w};
        float pe[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pe[e] = (a.pen && ib0 + e < a.I) ? a.pen[ib0 + e] : 0.f;
        float gr[8];
        if (divN.pow2 && divM.pow2) {          // wave-uniform: exact scaling instead of 8-16 IEEE divisions
#pragma unroll
            for (int e = 0; e < 8; ++e) gr[e] = (a.form == 0) ? grad_norm1(pv[e], a.N, divN.inv, true)
                                                              : grad_norm2(pv[e], nv[e], a.N, a.M, divN.inv, divM.inv, true);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) gr[e] = (a.form == 0) ? grad_norm1(pv[e], a.N, 0.f, false)
                                                              : grad_norm2(pv[e], nv[e], a.N, a.M, 0.f, 0.f, false);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            apply_w_update(gr[e], pe[e], a.l2, a.lr, a.mom, wv[e], dv[e]);
            wt[n][e] = wv[e];
        }
        if (side.vec8) {
            stream_store4(a.W + o, wv[0], wv[1], wv[2], wv[3]);
            stream_store4(a.W + o + 4, wv[4], wv[5], wv[6], wv[7]);
            stream_store4(a.dW + o, dv[0], dv[1], dv[2], dv[3]);
            stream_store4(a.dW + o + 4, dv[4], dv[5], dv[6], dv[7]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (ib0 + e < a.I) { a.W[o + e] = wv[e]; a.dW[o + e] = dv[e]; }
        }
    }
    if (a.fused && a.Wt) {                            // maintained transpose (prop-down P operand)
        const int ja = side.jb[0];                    // NJ == 2: the lane's two columns are adjacent: ja, ja + 1
        const bool pair = NJ == 2 && jok[0] && jok[NJ - 1] && ((a.ldwt & 1) == 0);     // (ja is even)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (ib0 + e >= a.I) break;
            float *dst = a.Wt + (size_t)(ib0 + e) * a.ldwt + ja;
            if (pair) {
                stream_store2(dst, wt[0][e], wt[NJ - 1][e]);
            } else {
                if (jok[0]) dst[0] = wt[0][e];
                if (NJ == 2 && jok[NJ - 1]) dst[1] = wt[NJ - 1][e];
            }
        }
    }
    BM_GSTAMP(2);
}

// split path: W update from (all-reduced) raw sums
struct ApplyWArgs {
    const float *raw, *raw2;
    float *W, *dW, *Wt;
    const float *pen;
    int I, J, ldw, ldwt;
    int form;
    float N, M, l2, lr, mom;
};
__global__ void apply_w_kernel(ApplyWArgs a) {
    const size_t n = (size_t)a.I * a.J;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e % (size_t)a.I), j = (int)(e / (size_t)a.I);
        const size_t o = (size_t)j * a.ldw + i;
        const float gr = (a.form == 0) ? grad_norm1(a.raw[o], a.N, 0.f, false) : grad_norm2(a.raw[o], a.raw2[o], a.N, a.M, 0.f, 0.f, false);
        float wv = a.W[o], dv = a.dW[o];
        apply_w_update(gr, a.pen ? a.pen[i] : 0.f, a.l2, a.lr, a.mom, wv, dv);
        a.W[o] = wv;
        a.dW[o] = dv;
        if (a.Wt) a.Wt[(size_t)i * a.ldwt + j] = wv;
    }
}

// Tiled form of the split apply (data-parallel step): 64 x 64 tiles of W, every global access a
// 16-byte one, the transpose for Wt staged through LDS.  The trailing `nbias` workgroups run
// rbm_bias_kernel's update (legal in the same launch when the W update does not need the
// penalty they produce, i.e. sparsity_cost == 0: `a.pen` is then null).
__device__ __forceinline__ void rbm_bias_update(const RbmBiasArgs &a, int c) {
    if (c < a.V) {
        const float g = a.sv[c] / a.N;
        const float d = a.lr * (a.mom * a.dvb[c] + g);
        a.dvb[c] = d;
        a.vb[c] = a.vb[c] + d;
    } else if (c < a.V + a.H) {
        const int h = c - a.V;
        const float qn = a.damping * a.q[h] + (1.0f - a.damping) * a.sq[h];
        a.q[h] = qn;
        const float pen = a.cost * (qn - a.target);
        a.pen[h] = pen;
        float g = a.sh[h] / a.N;
        g = g - pen;
        const float d = a.lr * (a.mom * a.dhb[h] + g);
        a.dhb[h] = d;
        a.hb[h] = a.hb[h] + d;
    }
}
__global__ __launch_bounds__(256) void apply_w_tiled_kernel(ApplyWArgs a, RbmBiasArgs b, int nbias) {
    __shared__ float tile[64][65];
    const int ntile = (int)gridDim.x - nbias;
    if ((int)blockIdx.x >= ntile) {
        rbm_bias_update(b, ((int)blockIdx.x - ntile) * 256 + (int)threadIdx.x);
        return;
    }
    const int tiles_i = (a.I + 63) / 64;
    const int i0 = ((int)blockIdx.x % tiles_i) * 64, j0 = ((int)blockIdx.x / tiles_i) * 64;
    const int tid = threadIdx.x, c4 = tid & 15, r0 = tid >> 4;       // 16 float4 per row, 16 rows per pass
    const bool pow2 = grad_pow2(a.N) && grad_pow2(a.M);
    const float invN = 1.0f / a.N, invM = 1.0f / a.M;
    const int i = i0 + 4 * c4;
    float4 pe = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.pen && i < a.I) pe = *reinterpret_cast<const float4 *>(a.pen + i);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int jl = r0 + 16 * p, j = j0 + jl;
        if (j < a.J && i < a.I) {                                   // I % 4 == 0: whole float4 in range
            const size_t o = (size_t)j * a.ldw + i;
            const float4 r = *reinterpret_cast<const float4 *>(a.raw + o);
            float4 wv = *reinterpret_cast<const float4 *>(a.W + o), dv = *reinterpret_cast<const float4 *>(a.dW + o);
            float4 g;
            if (a.form == 0) {
                g = make_float4(grad_norm1(r.x, a.N, invN, pow2), grad_norm1(r.y, a.N, invN, pow2),
                                grad_norm1(r.z, a.N, invN, pow2), grad_norm1(r.w, a.N, invN, pow2));
            } else {
                const float4 r2 = *reinterpret_cast<const float4 *>(a.raw2 + o);
                g = make_float4(grad_norm2(r.x, r2.x, a.N, a.M, invN, invM, pow2), grad_norm2(r.y, r2.y, a.N, a.M, invN, invM, pow2),
                                grad_norm2(r.z, r2.z, a.N, a.M, invN, invM, pow2), grad_norm2(r.w, r2.w, a.N, a.M, invN, invM, pow2));
            }
            apply_w_update(g.x, pe.x, a.l2, a.lr, a.mom, wv.x, dv.x);
            apply_w_update(g.y, pe.y, a.l2, a.lr, a.mom, wv.y, dv.y);
            apply_w_update(g.z, pe.z, a.l2, a.lr, a.mom, wv.z, dv.z);
            apply_w_update(g.w, pe.w, a.l2, a.lr, a.mom, wv.w, dv.w);
            *reinterpret_cast<float4 *>(a.W + o) = wv;
            *reinterpret_cast<float4 *>(a.dW + o) = dv;
            tile[jl][4 * c4] = wv.x; tile[jl][4 * c4 + 1] = wv.y; tile[jl][4 * c4 + 2] = wv.z; tile[jl][4 * c4 + 3] = wv.w;
        }
    }
    if (!a.Wt) return;
    __syncthreads();
    const int jq = j0 + 4 * c4;                                     // 4 consecutive j of row i
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int il = r0 + 16 * p, ii = i0 + il;
        if (ii >= a.I || jq >= a.J) continue;
        float *dst = a.Wt + (size_t)ii * a.ldwt + jq;
        if (jq + 3 < a.J) {
            *reinterpret_cast<float4 *>(dst) = make_float4(tile[4 * c4][il], tile[4 * c4 + 1][il], tile[4 * c4 + 2][il], tile[4 * c4 + 3][il]);
        } else {
            for (int e = 0; e < 4; ++e) if (jq + e < a.J) dst[e] = tile[4 * c4 + e][il];
        }
    }
}
// host: tiled path needs whole 16-byte groups (I % 4 == 0, pitches % 4 == 0); else the elementwise kernels
static inline void launch_apply_w(const ApplyWArgs &a, const RbmBiasArgs *bias, hipStream_t st) {
    const bool tiled = (a.I % 4 == 0) && (a.ldw % 4 == 0) && (!a.Wt || a.ldwt % 4 == 0);
    if (tiled) {
        RbmBiasArgs b;
        memset(&b, 0, sizeof(b));
        int nb = 0;
        if (bias) { b = *bias; nb = (b.V + b.H + 255) / 256; }
        const int ntile = ((a.I + 63) / 64) * ((a.J + 63) / 64);
        hipLaunchKernelGGL(apply_w_tiled_kernel, dim3(ntile + nb), dim3(256), 0, st, a, b, nb);
    } else {
        if (bias) hipLaunchKernelGGL(rbm_bias_kernel, dim3((bias->V + bias->H + 255) / 256), dim3(256), 0, st, *bias);
        hipLaunchKernelGGL(apply_w_kernel, dim3(1024), dim3(256), 0, st, a);
    }
}

// ------------------------------------------------------ MultinomialLayer (layers.py:54-70)
// One wave per row of logits L[j][0..I) (written by act_kernel kind 3), in place:
//   means = M * softmax(l);  states = counts of M categorical draws (or = means when !sample).
// Same operation sequence as oracle/bm_oracle.c softmax_multinomial_row (bit-exact):
//   mx = max l;  e[i] = exp_neg(min(mx - l[i], 80));  c[i] = c[i-1] + e[i] SEQUENTIALLY (lane 0);
//   S = c[I-1];  means[i] = M * (e[i] / S);  draw d: t = u(row*M + d) * S, category = first c[i] > t.
// LDS: c[I] | e[I] (e is reused for the integer counts); I <= 8192.
struct SmArgs {
    float *L; int ld, I, J, M, sample;
    float *states, *negmeans;        // may be null; pitch ld_states (0: the pitch of L)
    int ld_states;
    PhiloxKey key; long long row0;
    // DBM sweeps (a Multinomial layer inside the stack): mean-field residual max|means - prev| -> atomicMax on
    // float bits, and the device-side "loop finished" flag of the mean-field loop (launch becomes a no-op)
    const float *prev; int ld_prev; unsigned *maxdiff;
    const int *skip;
};
__global__ __launch_bounds__(64) void softmax_multinomial_kernel(SmArgs a) {
    extern __shared__ float sm_dyn[];
    float *c = sm_dyn, *e = sm_dyn + a.I;
    const int row = blockIdx.x, lane = threadIdx.x;
    if (a.skip && *a.skip) return;
    float *l = a.L + (size_t)row * a.ld;
    const int lds = a.ld_states ? a.ld_states : a.ld;
    float mx = -3.402823466e38f;
    for (int i = lane; i < a.I; i += 64) mx = fmaxf(mx, l[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    for (int i = lane; i < a.I; i += 64) {
        float d = mx - l[i];
        if (d > 80.0f) d = 80.0f;
        e[i] = exp_neg(d);
    }
    __syncthreads();
    if (lane == 0) {                                 // canonical (sequential) prefix sums
        float run = 0.0f;
        int i = 0;
        for (; i + 8 <= a.I; i += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = e[i + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) { run = run + t[u]; c[i + u] = run; }
        }
        for (; i < a.I; ++i) { run = run + e[i]; c[i] = run; }
    }
    __syncthreads();
    const float S = c[a.I - 1], Mf = (float)a.M;
    float dmax = 0.f;
    for (int i = lane; i < a.I; i += 64) {
        const float m = Mf * (e[i] / S);
        if (a.prev) dmax = fmaxf(dmax, fabsf(m - a.prev[(size_t)row * a.ld_prev + i]));
        l[i] = m;
        if (a.negmeans) a.negmeans[(size_t)row * lds + i] = -m;
        if (a.states && !a.sample) a.states[(size_t)row * lds + i] = m;
    }
    if (a.maxdiff) {            // wave-uniform
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));
        if (lane == 0 && dmax > 0.f) atomicMax(a.maxdiff, __float_as_uint(dmax));
    }
    if (!a.states || !a.sample) return;
    __syncthreads();
    int *cnt = reinterpret_cast<int *>(e);
    for (int i = lane; i < a.I; i += 64) cnt[i] = 0;
    __syncthreads();
    for (int d = lane; d < a.M; d += 64) {
        const float u = philox_uniform_at(a.key, (unsigned long long)(a.row0 + row) * (unsigned long long)a.M + (unsigned long long)d);
        const float t = u * S;
        int lo = 0, hi = a.I - 1;                    // smallest i with c[i] > t (exists: t < S)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (c[mid] > t) hi = mid; else lo = mid + 1;
        }
        atomicAdd(cnt + lo, 1);
    }
    __syncthreads();
    for (int i = lane; i < a.I; i += 64) a.states[(size_t)row * lds + i] = (float)cnt[i];
}

// h_hat ~ Multinomial(M, uniform over K) (rbm.py:58): counts of floor(u * K); three independent
// vectors (streams t = 0, 1, 2: free_energy_op, F(x) and F(x~) of the PLL), hhat [3][K] zeroed by the caller
__global__ void mn_hhat_kernel(float *hhat, int K, int M, PhiloxKey k0, PhiloxKey k1, PhiloxKey k2) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= M) return;
    const PhiloxKey keys[3] = {k0, k1, k2};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        int idx = (int)(philox_uniform_at(keys[t], (unsigned long long)d) * (float)K);
        if (idx > K - 1) idx = K - 1;
        atomicAdd(hhat + (size_t)t * K + idx, 1.0f);
    }
}

// ----------------------------------------------------------------- elementwise
// tf.nn.dropout(x, keep): x / keep * floor(keep + u)   (base_rbm.py:417-418)
// X [rows][cols] pitch ldx -> Y pitch ldy; RNG index = flat0 + row*cols + col
__global__ void dropout_kernel(const float *X, int ldx, float *Y, int ldy, int rows, int cols, float keep,
                               PhiloxKey key, unsigned long long flat0) {
    const size_t n = (size_t)rows * cols;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / (size_t)cols, c = e % (size_t)cols;
        const float u = philox_uniform_at(key, flat0 + e);
        Y[r * ldy + c] = (X[r * ldx + c] / keep) * floorf(keep + u);
    }
}

// GaussianRBM placeholder: X / sigma (rbm.py:107)
__global__ void div_cols_kernel(const float *X, int ldx, const float *sigma, float *Y, int ldy, int rows, int cols) {
    const size_t n = (size_t)rows * cols;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / (size_t)cols, c = e % (size_t)cols;
        Y[r * ldy + c] = X[r * ldx + c] / sigma[c];
    }
}

// strided 2-D copy (dense user buffer <-> padded internal matrix)
__global__ void copy2d_kernel(const float *X, int ldx, float *Y, int ldy, int rows, int cols) {
    const size_t n = (size_t)rows * cols;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / (size_t)cols, c = e % (size_t)cols;
        Y[r * ldy + c] = X[r * ldx + c];
    }
}

// T[c][r] = A[r][c], 32 x 32 tiles through LDS (full-line reads and writes)
__global__ __launch_bounds__(256) void transpose_kernel(const float *A, int lda, float *T, int ldt, int rows, int cols) {
    __shared__ float t[32][33];
    const int tiles_c = (cols + 31) / 32;
    const int r0 = ((int)blockIdx.x / tiles_c) * 32, c0 = ((int)blockIdx.x % tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (r0 + r < rows && c0 + tx < cols) t[r][tx] = A[(size_t)(r0 + r) * lda + c0 + tx];
    __syncthreads();
    for (int c = ty; c < 32; c += 8)
        if (c0 + c < cols && r0 + tx < rows) T[(size_t)(c0 + c) * ldt + r0 + tx] = t[tx][c];
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// sum of (A-B)^2 (B may be null) over a [rows][cols] window into a double accumulator
// (msre :486-488; l2 :482-484)
__global__ void sqdiff_kernel(const float *A, int lda, const float *B, int ldb, int rows, int cols, double *out) {
    const size_t n = (size_t)rows * cols;
    double s = 0.0;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / (size_t)cols, c = e % (size_t)cols;
        const float d = B ? (A[r * lda + c] - B[r * ldb + c]) : A[r * lda + c];
        s += (double)d * (double)d;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

// The metric fetch of a training iteration (base_rbm.py:482-517) as few launches as its dependences allow (round 4: it
// was two memsets, two sqdiff launches and the index kernel, then the two free-energy kernels and a device-to-host copy -
// eight stream operations, ~120 us next to a 64 us update):
//   metrics_prep_kernel  zeroes the six double sums and the row accumulators and draws the PLL flip column of every row;
//   sqdiff2_kernel       both squared sums (msre over [B][V], l2 over W) in one grid, row-wise (no 64-bit divisions);
//   scal_to_host_kernel  writes the six sums into the caller's pinned ring (a 48-byte store over PCIe instead of a copy
//                        engine operation that the compute queue waits for).
struct MetricsPrepArgs {
    double *scal; float *rowacc; int n_rowacc; int *flip; int B, V; PhiloxKey key; unsigned long long row0;
};
__global__ __launch_bounds__(256) void metrics_prep_kernel(MetricsPrepArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    if (t < 6) a.scal[t] = 0.0;
    for (int e = t; e < a.n_rowacc; e += nt) a.rowacc[e] = 0.f;
    for (int b = t; b < a.B; b += nt) {                 // pll_rand = tf.random_uniform([B], 0, V, int32) (base_rbm.py:500-501)
        a.flip[b] = pll_flip_col(a.key, a.row0 + b, a.V);
    }
}
struct SqJob { const float *A; int lda; const float *B; int ldb; int rows, cols; double *out; };
__device__ __forceinline__ void sqdiff2_body(const SqJob &j0, const SqJob &j1, int blk, int nblk) {
    // (round 5: this kernel took 19.5 us of a 118 us metrics iteration - 256 waves walking 784 + 512 rows with ONE dependent
    //  load per trip.  Now 1024+ waves, four 16-byte loads of a row in flight per lane where the pitch allows it.)
    const int half = nblk >> 1;
    const bool second = blk >= half;
    const SqJob &j = second ? j1 : j0;
    const int nb = second ? nblk - half : half, b = second ? blk - half : blk;
    const int lane = threadIdx.x & 63, wv = b * (blockDim.x >> 6) + (threadIdx.x >> 6), nwv = nb * (blockDim.x >> 6);
    double s = 0.0;
    const bool vec = (j.lda & 3) == 0 && (((uintptr_t)j.A) & 15u) == 0 && (!j.B || ((j.ldb & 3) == 0 && (((uintptr_t)j.B) & 15u) == 0));
    for (int r = wv; r < j.rows; r += nwv) {
        const float *pa = j.A + (size_t)r * j.lda, *pb = j.B ? j.B + (size_t)r * j.ldb : nullptr;
        int c = 0;
        if (vec) {
            const int c4n = j.cols >> 2;                     // whole 16-byte groups of the row
            for (int g0 = 0; g0 < c4n; g0 += 256) {          // 4 groups per lane and trip, all loads before any use
                float4 x[4], y[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int g = g0 + u * 64 + lane;
                    const int gc = g < c4n ? g : c4n - 1;
                    x[u] = *reinterpret_cast<const float4 *>(pa + 4 * gc);
                    y[u] = pb ? *reinterpret_cast<const float4 *>(pb + 4 * gc) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (g0 + u * 64 + lane < c4n) {
                        const float d0 = x[u].x - y[u].x, d1 = x[u].y - y[u].y, d2 = x[u].z - y[u].z, d3 = x[u].w - y[u].w;
                        s += (double)d0 * (double)d0 + (double)d1 * (double)d1 + (double)d2 * (double)d2 + (double)d3 * (double)d3;
                    }
                }
            }
            c = 4 * c4n;
        }
        for (c += lane; c < j.cols; c += 64) {
            const float d = pb ? (pa[c] - pb[c]) : pa[c];
            s += (double)d * (double)d;
        }
    }
    // ONE atomic per workgroup: same-address double atomics cost ~16 ns each (a wave-level version spent 17 us on 1024)
    __shared__ double s_part[4];
    s = wave_sum(s);
    if (lane == 0) s_part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0 && j.rows > 0) atomicAdd(j.out, (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
}
__global__ __launch_bounds__(256) void sqdiff2_kernel(SqJob j0, SqJob j1) { sqdiff2_body(j0, j1, (int)blockIdx.x, (int)gridDim.x); }
__global__ void scal_to_host_kernel(const double *scal, double *dst) {
    if (threadIdx.x < 6) dst[threadIdx.x] = scal[threadIdx.x];
}

// ------------------------------------------------ free-energy hidden term (K5/K6)
// rowacc[j]  += sum_i softplus(z[j][i] + hb[i])           over this block's i-range
// rowacc2[j] += the same for the PLL-corrupted row x~ (one flipped column per row,
//               base_rbm.py:496-509):  z~ = z + (1 - 2 x_f) W[f][:]
struct FeArgs {
    Operand P, Q; int K;     // P = W (KM), Q = X rows (XM)
    int I, J;
    const float *hb;
    float *rowacc;           // [J], zero-initialised
    float *rowacc2;          // [J] or null
    const int *flip_col;     // [J] or null
    // MultinomialRBM free energy (rbm.py:52-62): the hidden term is (xW).h_hat instead of the softplus
    // sum; hvec [3][I] = the h_hat of free_energy_op, of F(x) and of F(x~); rowacc3 [J] takes F(x)'s
    const float *hvec;
    float *rowacc3;
};
template <bool FAST>
__global__ __launch_bounds__(NT, 1) void fe_hidden_kernel(FeArgs a) {
    using G = GeoAct;
    constexpr int TI = G::TI, TJ = G::TJ;
    __shared__ __attribute__((aligned(16))) float smem[G::SMEM_FLOATS];
    const int tiles_j = (a.J + TJ - 1) / TJ;
    int ti, tj;
    block_to_tile(tiles_j, ti, tj);
    const int i0 = ti * TI, j0 = tj * TJ;
    f32x4 acc[2][1];
    acc[0][0] = acc[1][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    NoSide none;
    KRange kr;
    kr.P1 = a.P; kr.Q1 = a.Q; kr.K1 = a.K;
    kr.P2 = a.P; kr.Q2 = a.Q; kr.K2 = 0;
    mainloop<XM, G, FAST, false>(acc, kr, i0, j0, smem, none);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w & 1, wj = w >> 1;
    const int g = lane >> 4, l15 = lane & 15;
    const int j = j0 + wj * 16 + l15;
    float s = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < a.J) {
        float delta = 0.f; int fc = -1;
        if (a.flip_col) {
            fc = a.flip_col[j];
            delta = 1.0f - 2.0f * a.Q.ptr[(size_t)j * a.Q.ld + fc];
        }
        float zz[8];
        lane_outputs<G>(acc, 0, zz);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i = i0 + wi * 32 + g * 8 + e;
            if (i < a.I) {
                if (a.hvec) {
                    s += zz[e] * a.hvec[i];
                    s3 += zz[e] * a.hvec[a.I + i];
                    if (fc >= 0) s2 += (zz[e] + delta * a.P.ptr[(size_t)fc * a.P.ld + i]) * a.hvec[2 * (size_t)a.I + i];
                } else {
                    const float z = zz[e] + a.hb[i];
                    s += softplus(z);
                    if (fc >= 0) s2 += softplus(z + delta * a.P.ptr[(size_t)fc * a.P.ld + i]);
                }
            }
        }
    }
    s += __shfl_xor(s, 16);  s += __shfl_xor(s, 32);
    s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
    s3 += __shfl_xor(s3, 16); s3 += __shfl_xor(s3, 32);
    if (g == 0 && j < a.J) {
        atomicAdd(a.rowacc + j, s);
        if (a.rowacc2) atomicAdd(a.rowacc2 + j, s2);
        if (a.rowacc3) atomicAdd(a.rowacc3 + j, s3);
    }
}

// visible term + batch sums: one wave per row.  Bernoulli: -x.vb (rbm.py:18);
// Gaussian: 0.5*||x - vb/sigma||^2 (rbm.py:110-112).  out[0] += F(x_j), out[1] += F(x~_j)
struct FeRowArgs {
    const float *X; int ld, V, B;
    const float *vb, *sigma;     // sigma null => Bernoulli
    const float *rowacc, *rowacc2, *rowacc3;    // rowacc3: MultinomialRBM's second F(x) (out[2]) or null
    const int *flip_col;
    double *out;
    // nslot > 0: rowacc / rowacc2 are slot partials [B][ld_part] (row major) left by the h0 pass (ActArgs::fe_flip): the hidden
    // term of row j is their sum over the slots in a fixed order
    int nslot, ld_part;
    // flip_col == null and has_key: the flip column of row j is computed here, pll_flip_col(key, row0 + j, V) (fused fetch)
    int has_key; PhiloxKey key; unsigned long long row0;
};
constexpr int FE_ROWS_PER_WG = 8;       // 4 waves x 2 rows (round 5: 16 rows per workgroup left 32 workgroups walking four rows each
                                        // with dependent loads - 21.9 us; fewer rows mean more same-address double atomics at ~16 ns)
__device__ __forceinline__ void fe_row_body(const FeRowArgs &a, int blk) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int q = 0; q < FE_ROWS_PER_WG / 4; ++q) {
        const int row = blk * FE_ROWS_PER_WG + q * 4 + wv;
        if (row >= a.B) break;                       // wave-uniform
        const float *x = a.X + (size_t)row * a.ld;
        const int fc = a.flip_col ? a.flip_col[row] : (a.has_key ? pll_flip_col(a.key, a.row0 + row, a.V) : -1);
        double t = 0.0, t2 = 0.0;
        int c = lane;
        if (!a.sigma && (a.ld & 3) == 0 && (((uintptr_t)a.X | (uintptr_t)a.vb) & 15u) == 0) {
            // Bernoulli visible term, 16-byte loads, all of a row's loads in flight before the first use (the scalar loop below
            // walked 13 dependent trips per row: most of this kernel's 12 us)
            const int c4n = a.V >> 2;
            for (int g0 = 0; g0 < c4n; g0 += 256) {
                float4 xv[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int g = g0 + u * 64 + lane, gc = g < c4n ? g : c4n - 1;
                    xv[u] = *reinterpret_cast<const float4 *>(x + 4 * gc);
                    bv[u] = *reinterpret_cast<const float4 *>(a.vb + 4 * gc);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int g = g0 + u * 64 + lane;
                    if (g < c4n) {
                        const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, bs4[4] = {bv[u].x, bv[u].y, bv[u].z, bv[u].w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float xf = (4 * g + r == fc) ? 1.0f - xs[r] : xs[r];
                            t -= (double)(xs[r] * bs4[r]);
                            t2 -= (double)(xf * bs4[r]);
                        }
                    }
                }
            }
            c = 4 * c4n + lane;
        }
        for (; c < a.V; c += 64) {
            const float xv = x[c];
            const float xf = (c == fc) ? 1.0f - xv : xv;
            if (a.sigma) {
                const float mu = a.vb[c] / a.sigma[c];
                t += 0.5 * (double)((xv - mu) * (xv - mu));
                t2 += 0.5 * (double)((xf - mu) * (xf - mu));
            } else {
                t -= (double)(xv * a.vb[c]);
                t2 -= (double)(xf * a.vb[c]);
            }
        }
        t = wave_sum(t);
        t2 = wave_sum(t2);
        double h1 = 0.0, h2 = 0.0;
        if (a.nslot > 0) {
            for (int q = lane; q < a.nslot; q += 64) {
                h1 += (double)a.rowacc[(size_t)row * a.ld_part + q];
                h2 += (double)a.rowacc2[(size_t)row * a.ld_part + q];
            }
            h1 = wave_sum(h1); h2 = wave_sum(h2);            // (a fixed tree: the same bits every run)
        } else {
            h1 = (double)a.rowacc[row];
            if (a.rowacc2) h2 = (double)a.rowacc2[row];
        }
        s0 += t - h1;
        if (a.rowacc2) s1 += t2 - h2;
        if (a.rowacc3) s2 += t - (double)a.rowacc3[row];
    }
    __shared__ double s_part[3][4];
    if (lane == 0) { s_part[0][wv] = s0; s_part[1][wv] = s1; s_part[2][wv] = s2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double v = (s_part[threadIdx.x][0] + s_part[threadIdx.x][1]) + (s_part[threadIdx.x][2] + s_part[threadIdx.x][3]);
        if (threadIdx.x == 0 || (threadIdx.x == 1 && a.rowacc2) || (threadIdx.x == 2 && a.rowacc3)) atomicAdd(a.out + threadIdx.x, v);
    }
}
__global__ __launch_bounds__(256) void fe_row_kernel(FeRowArgs a) { fe_row_body(a, (int)blockIdx.x); }
// the tail of a fused metric fetch in ONE launch: blocks [0, nb_sq) the two squared sums, the rest the free-energy rows
__global__ __launch_bounds__(256) void metrics_tail_kernel(SqJob j0, SqJob j1, FeRowArgs r, int nb_sq) {
    if ((int)blockIdx.x < nb_sq) sqdiff2_body(j0, j1, (int)blockIdx.x, nb_sq);     // workgroup-uniform
    else fe_row_body(r, (int)blockIdx.x - nb_sq);
}

// pll_rand = tf.random_uniform([B], 0, V, int32): minval + u32 % range (base_rbm.py:500-501)
__global__ void pll_index_kernel(int *out, int B, int V, PhiloxKey key, unsigned long long row0) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned long long idx = row0 + b;
    uint32_t w[4];
    philox_block(key, idx >> 2, w);
    out[b] = (int)(w[idx & 3] % (uint32_t)V);
}

// ------------------------------------------------------------------ DBM update
// Bias / running-mean / sparsity update of one DBM layer from raw column sums
// (dbm.py:550-590, 597-600, 611-615), including the reference's scalar-index quirk:
// the EMA input is the column sum of UNIT `layer` of that layer, broadcast to all units.
struct DbmBiasArgs {
    const float *s_pos, *s_neg;      // [n] column sums of mu_i (or X) and of H_i (or v)
    float *b, *db;                   // bias and its momentum buffer
    float *q, *mm, *pen;             // q_means, mu_means, penalty out (null for the visible layer)
    int n, layer;
    float N, M, lr, mom, damping, cost, target;
};
__device__ __forceinline__ void dbm_bias_update(const DbmBiasArgs &a, int c) {
    float g = a.s_pos[c] / a.N - a.s_neg[c] / a.M;           // reduce_mean(mu) - reduce_mean(H)   :553,573-576
    if (a.q) {
        const float qn = a.damping * a.q[c] + (1.0f - a.damping) * a.s_neg[a.layer];    // :582-584 (q_means[i] scalar)
        const float mn = a.damping * a.mm[c] + (1.0f - a.damping) * a.s_pos[a.layer];   // :585-587
        a.q[c] = qn;
        a.mm[c] = mn;
        const float p1 = a.cost * (qn - a.target);
        const float p2 = a.cost * (mn - a.target);
        const float pen = p1 + p2;                                                      // :588-589
        a.pen[c] = pen;
        g = g - pen;                                                                    // :591
    }
    const float d = a.lr * (a.mom * a.db[c] + g);
    a.db[c] = d;
    a.b[c] = a.b[c] + d;
}
__global__ void dbm_bias_kernel(DbmBiasArgs a) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < a.n) dbm_bias_update(a, c);
}
// every layer's bias update of one DBM update as ONE launch (blockIdx.y = job: the visible layer, then the hidden layers): the
// three launches of a 2-layer stack were 3 us of work each behind 4 - 8 us of launch latency at the tail of the update
constexpr int DBM_BIAS_JOBS = 1 + 4;              // 1 + BM_DBM_MAX_LAYERS
struct DbmBiasMulti { DbmBiasArgs job[DBM_BIAS_JOBS]; };
__global__ void dbm_bias_multi_kernel(DbmBiasMulti m) {
    const DbmBiasArgs &a = m.job[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < a.n) dbm_bias_update(a, c);
}

// Max-norm column rescale (dbm.py:511-513, :603-607):  W[:,c] *= min(||W[:,c]||, c_max) / max(||W[:,c]||, 1e-8)
// Two kernels.  maxnorm_kernel: one workgroup per 16 columns computes ||.||^2 as the canonical chain
// sum_j fma(w_j, w_j, acc) - the diagonal of the 16x16 Gram block the MFMA produces when both operands are the
// column block - and leaves min(norm, c_max) / max(norm, 1e-8) per column in `num` / `den`.
// maxnorm_scale_kernel: 32 x 32 tiles over the whole matrix rescale W in place and rewrite the maintained
// transpose through LDS, both with full-line accesses on every CU (the one-kernel form of round 1 wrote the
// transpose 4 bytes at a time from 32 - 64 workgroups: 30 us per matrix).
struct MaxNormArgs {
    float *W, *Wt;          // [J][I] pitch ldw, transpose [I][J] pitch ldwt
    int I, J, ldw, ldwt;
    float max_norm;
    float *norm_out;        // [I] column norms (W_norm metric) or null
    float *num, *den;       // [I] each: the column factors
    int c_first, c_end;     // the columns this launch works on ([0, I) unless a rank owns a column slice: bm_xchg.hip);
                            // c_first % 32 == 0, c_end % 32 == 0 or c_end == I
};
// maxnorm_kernel: a workgroup owns 16 columns, its wave 0 runs their chain (768 dependent MFMAs for 3072 rows = 10 us
// is the floor: the chain of a column is sequential over the rows); all four waves stream the rows
// through two LDS slots in chunks of 256, up to THREE chunks ahead in registers (what 19 GB/s per chain need at
// ~2.5 us of memory latency is 48 KiB in flight).  Every load is issued unconditionally from a clamped address and zeroed
// afterwards: loads under divergent branches are waited for one by one.  History at 3072 x 5000: 80 us (load and
// MFMA phases alternating, branchy loads), 95 / 70 us (64 columns per workgroup, one chunk ahead: latency bound),
// 34 us now.
constexpr int MN_CH = 256;                       // rows per chunk (512: 41 us against 34)
constexpr int MN_COLS = 16;                      // columns per workgroup
constexpr int MN_NV = MN_CH * (MN_COLS / 4) / NT;   // float4 per thread and chunk
struct MnRegs { float4 v[MN_NV]; };
__device__ __forceinline__ void maxnorm_load(const MaxNormArgs &a, int c0, int r0, bool full, MnRegs &r) {
    const int tid = threadIdx.x;
    if (full) {                                           // workgroup-uniform
#pragma unroll
        for (int n = 0; n < MN_NV; ++n) {
            const int f = tid + n * NT, row = r0 + (f >> 2), c = c0 + 4 * (f & 3);
            const int rc = row < a.J ? row : a.J - 1;
            r.v[n] = *reinterpret_cast<const float4 *>(a.W + (size_t)rc * a.ldw + c);
        }
    } else {
#pragma unroll
        for (int n = 0; n < MN_NV; ++n) {
            const int f = tid + n * NT, row = r0 + (f >> 2), c = c0 + 4 * (f & 3);
            const int rc = row < a.J ? row : a.J - 1;
            const float *src = a.W + (size_t)rc * a.ldw;
            const int last = a.I - 1;
            r.v[n] = make_float4(src[c < last ? c : last], src[c + 1 < last ? c + 1 : last], src[c + 2 < last ? c + 2 : last],
                                 src[c + 3 < last ? c + 3 : last]);
        }
    }
}
__device__ __forceinline__ void maxnorm_store(const MaxNormArgs &a, int c0, int r0, float *buf, MnRegs &r) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int n = 0; n < MN_NV; ++n) {
        const int f = tid + n * NT, row = f >> 2, c4 = f & 3, c = c0 + 4 * c4;
        const bool rok = r0 + row < a.J;
        float4 v = r.v[n];
        v.x = (rok && c < a.I) ? v.x : 0.f;
        v.y = (rok && c + 1 < a.I) ? v.y : 0.f;
        v.z = (rok && c + 2 < a.I) ? v.z : 0.f;
        v.w = (rok && c + 3 < a.I) ? v.w : 0.f;
        *reinterpret_cast<float4 *>(buf + row * 16 + 4 * c4) = v;
    }
}
__global__ __launch_bounds__(NT) void maxnorm_kernel(MaxNormArgs a) {
    static_assert(NT == 256 && MN_NV >= 1 && MN_NV * NT == MN_CH * (MN_COLS / 4), "loader split");
    __shared__ __attribute__((aligned(16))) float sA[2][MN_CH * 16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int c0 = a.c_first + blockIdx.x * MN_COLS;
    const bool full = (a.ldw & 3) == 0 && (((uintptr_t)a.W) & 15u) == 0 && c0 + MN_COLS <= a.I;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nch = (a.J + MN_CH - 1) / MN_CH;
    MnRegs R0, R1, R2;
    maxnorm_load(a, c0, 0, full, R0);
    if (nch > 1) maxnorm_load(a, c0, MN_CH, full, R1);
    if (nch > 2) maxnorm_load(a, c0, 2 * MN_CH, full, R2);
    // iteration ch: chunk ch's registers -> slot ch % 2, chunk ch + 3's loads issued, barrier, wave 0 runs chunk ch.
    // Slot ch % 2 is written again in iteration ch + 2: wave 0 finished reading it before the barrier of iteration ch + 1.
#define BM_MN_STEP(R, CH)                                                                         \
    if ((CH) < nch) {                                                                             \
        maxnorm_store(a, c0, (CH) * MN_CH, sA[(CH) & 1], R);                                      \
        if ((CH) + 3 < nch) maxnorm_load(a, c0, ((CH) + 3) * MN_CH, full, R);                     \
        wg_barrier();          /* not __syncthreads(): its fence would drain the chunks in flight */ \
        if (w == 0) {                                                                             \
            const float *img = sA[(CH) & 1];                                                      \
            _Pragma("unroll 2")                                                                   \
            for (int s = 0; s < MN_CH / 4; s += 8) {      /* rows beyond J are zero: fma(0, 0, acc) == acc */ \
                float d[8];                                                                       \
                _Pragma("unroll")                                                                 \
                for (int u = 0; u < 8; ++u) d[u] = img[(4 * (s + u) + g) * 16 + l15];             \
                _Pragma("unroll")                                                                 \
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(d[u], d[u], acc, 0, 0, 0); \
            }                                                                                     \
        }                                                                                         \
    }
    for (int ch = 0; ch < nch; ch += 3) {
        BM_MN_STEP(R0, ch)
        BM_MN_STEP(R1, ch + 1)
        BM_MN_STEP(R2, ch + 2)
    }
#undef BM_MN_STEP
    const int c = c0 + l15;
    if (w == 0 && (l15 >> 2) == g && c < a.I) {              // diagonal element i == j == l15
        const float nrm = sqrtf(acc[l15 & 3]);
        a.num[c] = fminf(nrm, a.max_norm);                   // tf.minimum(T_norm, max_norm)
        a.den[c] = fmaxf(nrm, 1e-8f);                        // tf.maximum(T_norm, 1e-8)
        if (a.norm_out) a.norm_out[c] = nrm;
    }
}
__global__ __launch_bounds__(256) void maxnorm_scale_kernel(MaxNormArgs a) {
    __shared__ float t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int tiles_c = (a.c_end - a.c_first + 31) / 32;
    const int c0 = a.c_first + (blockIdx.x % tiles_c) * 32, j0 = (blockIdx.x / tiles_c) * 32;
    const int c = c0 + tx;
    const float num = c < a.I ? a.num[c] : 0.f, den = c < a.I ? a.den[c] : 1.f;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r;
        float wn = 0.f;
        if (j < a.J && c < a.I) {
            const size_t o = (size_t)j * a.ldw + c;
            wn = (a.W[o] * num) / den;                       // T * min / max, left to right
            a.W[o] = wn;
        }
        t[r][tx] = wn;
    }
    if (!a.Wt) return;
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {                       // row r of the transposed tile = column c0 + r
        const int cc = c0 + r, j = j0 + tx;
        if (cc < a.I && j < a.J) a.Wt[(size_t)cc * a.ldwt + j] = t[tx][r];
    }
}

// device-side mean-field loop control (dbm.py:449-452): after each sweep, advance the counter
// and latch `done` when the residual no longer exceeds the tolerance; `init` evaluates the
// step-0 condition from the residual between the persistent mu and the init values.
// blk [nblk]: per-workgroup residuals of the sweep's act_kernel launches (read, then zeroed for the next sweep)
__global__ __launch_bounds__(256) void mf_ctl_kernel(MfCtl *c, float tol, int init, float *blk, int nblk) {
    __shared__ float s_m[4];
    float m = 0.f;
    for (int e = threadIdx.x; e < nblk; e += 256) { m = fmaxf(m, blk[e]); blk[e] = 0.f; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float resid = fmaxf(fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3])), __uint_as_float(c->maxdiff));
    if (init) {
        c->steps = 0;
        c->done = !(resid > tol);
    } else if (!c->done) {
        c->steps += 1;
        c->done = !(resid > tol);
    }
    c->maxdiff = 0u;
}

// Data-parallel form of mf_ctl_kernel, split around the all-reduce(max) of the residual over the ranks
// (the loop condition of dbm.py:449-452 is over ALL rows of the global minibatch):
//   mf_resid_kernel: local residual of the sweep -> c->resid (slots and the atomic cell are reset)
//   [ncclAllReduce(max) of c->resid on the same stream]
//   mf_latch_kernel: the counter / `done` update of mf_ctl_kernel from the reduced value
__global__ __launch_bounds__(256) void mf_resid_kernel(MfCtl *c, float *blk, int nblk) {
    __shared__ float s_m[4];
    float m = 0.f;
    for (int e = threadIdx.x; e < nblk; e += 256) { m = fmaxf(m, blk[e]); blk[e] = 0.f; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x != 0) return;
    c->resid = fmaxf(fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3])), __uint_as_float(c->maxdiff));
    c->maxdiff = 0u;
}
__global__ void mf_latch_kernel(MfCtl *c, float tol, int init) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (init) {
        c->steps = 0;
        c->done = !(c->resid > tol);
    } else if (!c->done) {
        c->steps += 1;
        c->done = !(c->resid > tol);
    }
}

// ||A - B||_inf over a [rows][cols] window -> atomicMax on float bits (mean-field cond, dbm.py:449-452).
// Rows over workgroups, columns over threads (coalesced), ONE atomic per workgroup: same-address atomics serialise
// in the L2 at ~12 ns each (2048 of them made this kernel take 25 us).
__global__ __launch_bounds__(256) void maxabsdiff_kernel(const float *A, int lda, const float *B, int ldb, int rows, int cols, unsigned *out) {
    __shared__ float s_m[4];
    float m = 0.f;
    const bool vec = ((lda | ldb) & 3) == 0 && ((((uintptr_t)A) | ((uintptr_t)B)) & 15u) == 0;
    if (vec) {
        // 16-byte loads, four pairs in flight per thread (one row pair per 256 threads took 33 us for 256 x 5000:
        // twenty dependent round trips)
        const int c4n = cols >> 2;
        const long long n = (long long)rows * c4n;
        const long long stride = (long long)gridDim.x * 256;
        long long e = (long long)blockIdx.x * 256 + threadIdx.x;
        for (; e + 3 * stride < n; e += 4 * stride) {
            float4 x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long f = e + u * stride;
                const int r = (int)(f / c4n), c = 4 * (int)(f % c4n);
                x[u] = *reinterpret_cast<const float4 *>(A + (size_t)r * lda + c);
                y[u] = *reinterpret_cast<const float4 *>(B + (size_t)r * ldb + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                m = fmaxf(m, fmaxf(fmaxf(fabsf(x[u].x - y[u].x), fabsf(x[u].y - y[u].y)), fmaxf(fabsf(x[u].z - y[u].z), fabsf(x[u].w - y[u].w))));
        }
        for (; e < n; e += stride) {
            const int r = (int)(e / c4n), c = 4 * (int)(e % c4n);
            const float4 x = *reinterpret_cast<const float4 *>(A + (size_t)r * lda + c), y = *reinterpret_cast<const float4 *>(B + (size_t)r * ldb + c);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(x.x - y.x), fabsf(x.y - y.y)), fmaxf(fabsf(x.z - y.z), fabsf(x.w - y.w))));
        }
        for (int r = blockIdx.x; r < rows; r += gridDim.x)           // the columns beyond the last whole quad
            for (int c = 4 * c4n + threadIdx.x; c < cols; c += 256) m = fmaxf(m, fabsf(A[(size_t)r * lda + c] - B[(size_t)r * ldb + c]));
    } else {
        for (int r = blockIdx.x; r < rows; r += gridDim.x) {
            const float *pa = A + (size_t)r * lda, *pb = B + (size_t)r * ldb;
            for (int c = threadIdx.x; c < cols; c += 256) m = fmaxf(m, fabsf(pa[c] - pb[c]));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        if (m > 0.f) atomicMax(out, __float_as_uint(m));
    }
}

// Approximate-inference init of the FIRST hidden layer from the hoisted chain (dbm.py:434-446 with mean_field()'s X.W0):
// out = sigmoid(mult * z + bmult * b) with z = the stored raw pre-activation - the arithmetic of act_epilogue, operation by
// operation, so the result is the bits of the GEMM pass it replaces - and the step-0 residual max |out - prev| of the loop
// condition (dbm.py:449-452) into this workgroup's slot (gridDim.x <= BM_MF_SLOTS) or one atomic.  Rows over workgroups.
__global__ __launch_bounds__(256) void mf_init0_kernel(const float *Z, int ldz, const float *bias, const float *prev, int ldp,
                                                       float *out, int ldo, int rows, int cols, float mult, float bmult, int lit,
                                                       unsigned *maxdiff, float *blk) {
    __shared__ float s_m[4];
    float dm = 0.f;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const float *z = Z + (size_t)r * ldz, *p = prev + (size_t)r * ldp;
        float *o = out + (size_t)r * ldo;
        for (int c = threadIdx.x; c < cols; c += 256) {
            const float x = mult * z[c];
            const float b = bmult * bias[c];
            const float m = lit ? sigmoid_literal(x + b) : sigmoid(x + b);
            dm = fmaxf(dm, fabsf(m - p[c]));
            o[c] = m;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dm = fmaxf(dm, __shfl_xor(dm, off));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = dm;
    __syncthreads();
    if (threadIdx.x == 0) {
        dm = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        if (blk && gridDim.x <= BM_MF_SLOTS) blk[blockIdx.x] = dm;
        else if (dm > 0.f) atomicMax(maxdiff, __float_as_uint(dm));
    }
}

// ------------------------------------------------------------- host launchers
template <class G> static inline int tile_grid(int I, int J) { return ((I + G::TI - 1) / G::TI) * ((J + G::TJ - 1) / G::TJ); }

template <class G, int MINB, int STG>
static inline void launch_act_geo(const ActArgs &a_in, hipStream_t st) {
    const ActArgs &a = a_in;
    TileMap tmap;
    {   // operand bytes one tile row (TI outputs along i) / one tile column (TJ rows) pulls through the L2
        const double kt = (double)a.K1 + (double)a.K2;
        tmap = make_tile_map((a.I + G::TI - 1) / G::TI, (a.J + G::TJ - 1) / G::TJ, kt * G::TI * 4.0, kt * G::TJ * 4.0, a.map_xi);
    }
    const bool seg2 = a.K2 > 0;
    const int pl = a.p_xm ? XM : KM;
    const bool fast = operand_fast(a.P1, pl, a.K1) && operand_fast(a.Q1, XM, a.K1) &&
                      (!seg2 || (operand_fast(a.P2, KM, a.K2) && operand_fast(a.Q2, XM, a.K2)));
    const dim3 grid(tile_grid<G>(a.I, a.J)), blk(G::NT);
    // (shapes without 16-byte loads have ONE flavour: every chunk passes through registers)
    if constexpr (G::MI == 1) {
        if (a.p_xm) {                       // x-major P: single segment only (RBM prop-down from W)
            if (fast) hipLaunchKernelGGL((act_kernel<G, MINB, false, true, 0, XM, STG>), grid, blk, 0, st, a, tmap);
            else      hipLaunchKernelGGL((act_kernel<G, MINB, false, false, 0, XM, STG_DMA>), grid, blk, 0, st, a, tmap);
            return;
        }
    }
    if (seg2) {
        if (fast) hipLaunchKernelGGL((act_kernel<G, MINB, true, true, 0, KM, STG>), grid, blk, 0, st, a, tmap);
        else      hipLaunchKernelGGL((act_kernel<G, MINB, true, false, 0, KM, STG_DMA>), grid, blk, 0, st, a, tmap);
    } else {
        if (fast) hipLaunchKernelGGL((act_kernel<G, MINB, false, true, 0, KM, STG>), grid, blk, 0, st, a, tmap);
        else      hipLaunchKernelGGL((act_kernel<G, MINB, false, false, 0, KM, STG_DMA>), grid, blk, 0, st, a, tmap);
    }
}

// the h0 pass of a fused metric fetch (ActArgs::fe_flip): the 8-wave 32 x 64 tile, LDS-DMA, slab order - a compile-time
// flavour of its own (FE) and not a tuner case, so that the kernels of the plain update carry none of its code (same-box A/B:
// a runtime branch in the shared epilogue cost the headline 0.5 us per update).  The 32 x 32 tiles measured within 1 us of it.
static inline void launch_act_fe(const ActArgs &a, hipStream_t st) {
    using G = GeoAct8;
    const double kt = (double)a.K1;
    const TileMap tmap = make_tile_map((a.I + G::TI - 1) / G::TI, (a.J + G::TJ - 1) / G::TJ, kt * G::TI * 4.0, kt * G::TJ * 4.0, -1);
    const bool fast = operand_fast(a.P1, a.p_xm ? XM : KM, a.K1) && operand_fast(a.Q1, XM, a.K1);
    const dim3 grid(tile_grid<G>(a.I, a.J)), blk(G::NT);
    if (a.p_xm) {
        if (fast) hipLaunchKernelGGL((act_kernel<G, 1, false, true, 0, XM, STG_DMA, true>), grid, blk, 0, st, a, tmap);
        else      hipLaunchKernelGGL((act_kernel<G, 1, false, false, 0, XM, STG_DMA, true>), grid, blk, 0, st, a, tmap);
    } else {
        if (fast) hipLaunchKernelGGL((act_kernel<G, 1, false, true, 0, KM, STG_DMA, true>), grid, blk, 0, st, a, tmap);
        else      hipLaunchKernelGGL((act_kernel<G, 1, false, false, 0, KM, STG_DMA, true>), grid, blk, 0, st, a, tmap);
    }
}

// "reference arithmetic" launches (ActArgs::lit, bm_dbm_set_sigmoid_literal): the literal float32 tf.sigmoid in the epilogue -
// a compile-time flavour (LIT) on ONE geometry (32 x 32 tiles, LDS-DMA, slab order; x-major P needs MI == 1): a parity mode,
// not a tuner case, so the kernels of the default path carry none of its code.  Same canonical accumulation order as
// every other geometry (bm_gemm.h), hence the same pre-activations bit for bit.
static inline void launch_act_lit(const ActArgs &a, hipStream_t st) {
    using G = GeoActS;
    const double kt = (double)a.K1 + (double)a.K2;
    const TileMap tmap = make_tile_map((a.I + G::TI - 1) / G::TI, (a.J + G::TJ - 1) / G::TJ, kt * G::TI * 4.0, kt * G::TJ * 4.0, -1);
    const bool seg2 = a.K2 > 0;
    const bool fast = operand_fast(a.P1, a.p_xm ? XM : KM, a.K1) && operand_fast(a.Q1, XM, a.K1) &&
                      (!seg2 || (operand_fast(a.P2, KM, a.K2) && operand_fast(a.Q2, XM, a.K2)));
    const dim3 grid(tile_grid<G>(a.I, a.J)), blk(G::NT);
    if (a.p_xm) {
        if (fast) hipLaunchKernelGGL((act_kernel<G, 2, false, true, 0, XM, STG_DMA, false, true>), grid, blk, 0, st, a, tmap);
        else      hipLaunchKernelGGL((act_kernel<G, 2, false, false, 0, XM, STG_DMA, false, true>), grid, blk, 0, st, a, tmap);
    } else if (seg2) {
        if (fast) hipLaunchKernelGGL((act_kernel<G, 2, true, true, 0, KM, STG_DMA, false, true>), grid, blk, 0, st, a, tmap);
        else      hipLaunchKernelGGL((act_kernel<G, 2, true, false, 0, KM, STG_DMA, false, true>), grid, blk, 0, st, a, tmap);
    } else {
        if (fast) hipLaunchKernelGGL((act_kernel<G, 2, false, true, 0, KM, STG_DMA, false, true>), grid, blk, 0, st, a, tmap);
        else      hipLaunchKernelGGL((act_kernel<G, 2, false, false, 0, KM, STG_DMA, false, true>), grid, blk, 0, st, a, tmap);
    }
}

// mean-field passes (ActArgs::prev / maxdiff / skip / chk_ctl / acc_init): the MF flavour (ActSide<.., MF>), LDS-DMA, slab order.
// Two tiles, by rule: 32 x 64 (8 waves, one workgroup per CU) where that gives every CU a tile, else 32 x 32 (4 waves) - what the
// launch tuner picked for these passes at 784-512-1024 x 512 (profiles/r5_dbm_kernel_stats.csv); BM355_DEBUG=mf_geo=8|1 forces
// one.  The literal-sigmoid mode (LIT) takes the 32 x 32 tile.
template <class G, int MINB, bool LIT>
static inline void launch_act_mf_geo(const ActArgs &a, hipStream_t st, unsigned dyn_lds = 0) {
    const double kt = (double)a.K1 + (double)a.K2;
    const TileMap tmap = make_tile_map((a.I + G::TI - 1) / G::TI, (a.J + G::TJ - 1) / G::TJ, kt * G::TI * 4.0, kt * G::TJ * 4.0, -1);
    const bool seg2 = a.K2 > 0;
    const bool fast = operand_fast(a.P1, a.p_xm ? XM : KM, a.K1) && operand_fast(a.Q1, XM, a.K1) &&
                      (!seg2 || (operand_fast(a.P2, KM, a.K2) && operand_fast(a.Q2, XM, a.K2)));
    const dim3 grid(tile_grid<G>(a.I, a.J)), blk(G::NT);
    if (a.p_xm) {
        if (fast) hipLaunchKernelGGL((act_kernel<G, MINB, false, true, 0, XM, STG_DMA, false, LIT, true>), grid, blk, dyn_lds, st, a, tmap);
        else      hipLaunchKernelGGL((act_kernel<G, MINB, false, false, 0, XM, STG_DMA, false, LIT, true>), grid, blk, dyn_lds, st, a, tmap);
    } else if (seg2) {
        if (fast) hipLaunchKernelGGL((act_kernel<G, MINB, true, true, 0, KM, STG_DMA, false, LIT, true>), grid, blk, dyn_lds, st, a, tmap);
        else      hipLaunchKernelGGL((act_kernel<G, MINB, true, false, 0, KM, STG_DMA, false, LIT, true>), grid, blk, dyn_lds, st, a, tmap);
    } else {
        if (fast) hipLaunchKernelGGL((act_kernel<G, MINB, false, true, 0, KM, STG_DMA, false, LIT, true>), grid, blk, dyn_lds, st, a, tmap);
        else      hipLaunchKernelGGL((act_kernel<G, MINB, false, false, 0, KM, STG_DMA, false, LIT, true>), grid, blk, dyn_lds, st, a, tmap);
    }
}
static inline void launch_act_mf(const ActArgs &a, hipStream_t st) {
    if (a.lit && a.kind == 0) { launch_act_mf_geo<GeoActS, 1, true>(a, st); return; }
    static const int force = bm::dbg("mf_geo") ? atoi(bm::dbg("mf_geo")) : 0;
    int dev = 0, ncu = 256;
    static int ncu_cached = 0;
    if (!ncu_cached) {
        hipDeviceProp_t pr;
        (void)hipGetDevice(&dev);
        ncu_cached = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    ncu = ncu_cached;
    const bool wide = force ? force == 8 : tile_grid<GeoAct8>(a.I, a.J) >= ncu;
    // (the 32 x 32 tile of this flavour takes 64 KiB for its ring + 16 KiB for the control words: one workgroup per CU)
    if (wide) launch_act_mf_geo<GeoAct8, 1, false>(a, st);
    else      launch_act_mf_geo<GeoActS, 1, false>(a, st);
}

// fast-binary launch (a.b3 filled).  Three tiles: 64 x 64 / 8 waves and 64 x 32 / 4 waves (one workgroup per CU: the
// ring takes most of the LDS), 32 x 64 / 4 waves with TWO workgroups per CU (80 KiB each: one workgroup's epilogue -
// sigmoid, draw, the AIS softplus terms - runs under the other's matrix work).  Every workgroup owns a strip of
// tile columns.  BM355_DEBUG=bf3_geo=8|4|2 forces one.
template <class G, int WGS_PER_CU>
static inline void launch_act_bf3_geo(const ActArgs &a, hipStream_t st) {
    static int ncu = 0;
    if (!ncu) { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); ncu = (hipGetDeviceProperties(&pr, d) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }
    Bf3Strip sp;
    sp.tiles_i = (a.I + G::TI - 1) / G::TI; sp.tiles_j = (a.J + G::TJ - 1) / G::TJ;
    static const int abl_env = bm::dbg("bf3_abl") ? atoi(bm::dbg("bf3_abl")) : 0;
    sp.abl = abl_env;
    sp.strips = (ncu * WGS_PER_CU) / sp.tiles_i;
    if (sp.strips < 1) sp.strips = 1;
    if (sp.strips > sp.tiles_j) sp.strips = sp.tiles_j;
    const dim3 grid(sp.tiles_i * sp.strips), blk(G::NT);
    constexpr int MINW = WGS_PER_CU * G::NW / 4 > 0 ? WGS_PER_CU * G::NW / 4 : 1;          // waves per SIMD the grid needs
    if (a.b3.K2 > 0) hipLaunchKernelGGL((act_bf3_kernel<G, true, MINW>), grid, blk, 0, st, a, sp);
    else             hipLaunchKernelGGL((act_bf3_kernel<G, false, MINW>), grid, blk, 0, st, a, sp);
}
static inline void launch_act_bf3_as(int geo, const ActArgs &a, hipStream_t st) {
    if (geo == 8)      launch_act_bf3_geo<GeoGrad8, 1>(a, st);
    else if (geo == 2) launch_act_bf3_geo<GeoBf3S, 2>(a, st);
    else               launch_act_bf3_geo<GeoAct, 1>(a, st);
}

// ---- act_kernel geometry choice ---------------------------------------------------------
// Four geometries compute bit-identical results (tests run all of them); which one is fastest
// depends on how the output tiles fill the 256 CUs and on the K length, and did not follow a
// simple rule in measurements (784x1024x512: 8-wave; AIS 20000 chains and 3072x5000: 32x32
// tiles with BK = 32, four workgroups per CU; DBM 784-512-1024 mean-field: 64x32).  So the
// launcher measures, ONCE per distinct shape and process, SYNCHRONOUSLY at the first launch of
// that shape: every candidate runs the caller's contraction on the caller's (read-only)
// operands with all OUTPUTS redirected to a scratch pool (so the tuning launches have no side
// effects: no double-counted row accumulators, no early write of a mean-field result), timed
// with one HIP event pair around TUNE_REP back-to-back launches, best of TUNE_ROUNDS rounds.
// After that the launch path is one table lookup: no event, no allocation, no synchronisation
// (round 1 rotated the candidates through the first 12 real launches, which put slower
// geometries and event markers into short timed runs).  BM355_DEBUG=act_geo=4|8|1|3 forces one
// geometry (experiments, tests); BM355_DEBUG=tune_log=1 prints the decisions.
static inline int act_geo_override() {
    static int v = -1;
    if (v < 0) { const char *e = bm::dbg("act_geo"); v = e ? atoi(e) : 0; }
    return v;
}
// geo: tile geometry 8 | 4 | 1 | 3, + 100 for register staging of the full chunks (default: LDS-DMA)
static inline void launch_act_as(int geo, const ActArgs &a, hipStream_t st) {
    // 208: 8 waves, DMA issued by waves 0-3 only (not a tuner candidate: within noise of 8 on every shape measured)
    if (geo == 208) { launch_act_geo<GeoAct8, 1, STG_DMAH>(a, st); return; }
    // 6: 64 x 64 tile, 8 waves of 32 x 16 (the outer-product geometry): half the operand traffic per flop of the
    // 32 x 64 tile, for outputs large enough to fill the chip with tiles of that size
    if (geo == 6 && !a.p_xm) { launch_act_geo<GeoGrad8, 1, STG_DMA>(a, st); return; }
    if (geo == 6) geo = 8;
    // 9: the 64 x 64 tile with BK = 32: 64 KiB LDS, two workgroups per CU (k-major P only)
    // (the second template argument is the kernel's waves per SIMD: 2 workgroups x 8 waves / 4 SIMDs)
    if (geo == 9 && !a.p_xm) { launch_act_geo<GeoGrad8h, 4, STG_DMA>(a, st); return; }
    if (geo == 9) geo = 8;
    // 5: 64 x 32 tile with BK = 32, three workgroups per CU (k-major P only)
    if (geo == 5 && !a.p_xm) { launch_act_geo<GeoAct32, 3, STG_DMA>(a, st); return; }
    // 7: the same with two workgroups per CU (256 registers per wave: the two-segment variant spills 140 bytes at 168)
    if (geo == 7 && !a.p_xm) { launch_act_geo<GeoAct32, 2, STG_DMA>(a, st); return; }
    if (geo == 5 || geo == 7) geo = 3;
    const bool reg = geo >= 100;
    geo %= 100;
    if (a.p_xm && geo == 4) geo = 8;        // x-major P exists for the MI == 1 geometries only
    if (reg) {
        if (geo == 8)      launch_act_geo<GeoAct8, 1, STG_REG>(a, st);
        else if (geo == 1) launch_act_geo<GeoActS, 2, STG_REG>(a, st);
        else if (geo == 3) launch_act_geo<GeoActS32, 4, STG_REG>(a, st);
        else               launch_act_geo<GeoAct, 1, STG_REG>(a, st);
    } else {
        if (geo == 8)      launch_act_geo<GeoAct8, 1, STG_DMA>(a, st);
        else if (geo == 1) launch_act_geo<GeoActS, 2, STG_DMA>(a, st);
        else if (geo == 3) launch_act_geo<GeoActS32, 4, STG_DMA>(a, st);
        else               launch_act_geo<GeoAct, 1, STG_DMA>(a, st);
    }
}
struct ActTune {
    static constexpr int NC = 12;
    int best = 0;
    int xi = 0;                  // XCD grid of the block -> tile map (TileMap), measured with the chosen geometry
    float t_us[NC] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
};
// scratch pool of the tuning launches (per process and device; grown on demand, never on the hot path)
struct TuneScratch {
    float *p = nullptr; size_t cap = 0; int dev = -1;
    float *get(size_t nfloats) {
        int d = 0; (void)hipGetDevice(&d);
        if (p && (d != dev || nfloats > cap)) { (void)hipFree(p); p = nullptr; cap = 0; }
        if (!p) {
            if (hipMalloc((void **)&p, nfloats * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return nullptr; }
            cap = nfloats; dev = d;
        }
        return p;
    }
};
// the caller's launch with every OUTPUT redirected into the scratch pool (false: no memory, keep the default choice)
static inline bool tune_redirect(const ActArgs &a, ActArgs &t) {
    static TuneScratch pool;
    const size_t mat = ((size_t)a.J * (size_t)a.ldo + 3) & ~(size_t)3;
    const size_t rowv = (((size_t)((a.I + 15) / 16) * (size_t)(a.ld_part > a.J ? a.ld_part : a.J)) + 3) & ~(size_t)3;   // slot partials
    const size_t sh16 = a.states16 ? ((size_t)a.J * (size_t)a.ld16 / 2 + 4) & ~(size_t)3 : 0;                           // bf16 shadow, in floats
    const size_t fe = a.fe_flip ? ((size_t)a.J * (size_t)a.fe_rm + 3) & ~(size_t)3 : 0;
    float *s = pool.get(3 * mat + 2 * rowv + BM_MF_SLOTS + 4 + sh16 + fe);
    if (!s) return false;
    t = a;
    if (a.fe_flip) t.fe_rowacc2 = s + 3 * mat + 2 * rowv + BM_MF_SLOTS + 4 + sh16;
    t.skip = nullptr;
    t.chk_ctl = nullptr;
    if (a.means) t.means = s;
    if (a.states) t.states = s + mat;
    if (a.negmeans) t.negmeans = s + 2 * mat;
    if (a.rowacc) t.rowacc = s + 3 * mat;        // (fe_flip: [J][fe_rm] <= rowv floats: fe_rm >= ceil(I/16), ld_part >= J)
    if (a.rowdot_out) t.rowdot_out = s + 3 * mat + rowv;
    if (a.maxdiff_blk) t.maxdiff_blk = s + 3 * mat + 2 * rowv;
    if (a.maxdiff) t.maxdiff = reinterpret_cast<unsigned *>(s + 3 * mat + 2 * rowv + BM_MF_SLOTS);
    if (a.states16) t.states16 = reinterpret_cast<uint16_t *>(s + 3 * mat + 2 * rowv + BM_MF_SLOTS + 4);
#ifdef BM_PROBE
    t.dbg = nullptr;
#endif
    return true;
}
static inline void tune_act_shape(const ActArgs &a, hipStream_t st, ActTune &T, long long flags) {
    static const int cand_geo[ActTune::NC] = {8, 4, 1, 3, 108, 104, 101, 103, 6, 5, 7, 9};
    constexpr int TUNE_REP = 4, TUNE_ROUNDS = 3;
    T.best = a.p_xm ? 8 : 4;
    ActArgs t;
    if (!tune_redirect(a, t)) return;                // no memory for the scratch outputs: keep the default
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); return; }
    float best_us[ActTune::NC] = {1e30f, 1e30f, 1e30f, 1e30f, 1e30f, 1e30f, 1e30f, 1e30f, 1e30f, 1e30f, 1e30f, 1e30f};
    for (int round = 0; round < TUNE_ROUNDS; ++round) {
        for (int c = 0; c < ActTune::NC; ++c) {
            if (a.p_xm && (cand_geo[c] % 100 == 4 || cand_geo[c] == 6 || cand_geo[c] == 5 || cand_geo[c] == 7 || cand_geo[c] == 9)) continue;   // not instantiated for an x-major P
            launch_act_as(cand_geo[c], t, st);                    // warm (instruction cache, clocks)
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < TUNE_REP; ++r) launch_act_as(cand_geo[c], t, st);
            (void)hipEventRecord(e1, st);
            if (hipEventSynchronize(e1) != hipSuccess) { (void)hipGetLastError(); continue; }
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && 1e3f * ms / TUNE_REP < best_us[c]) best_us[c] = 1e3f * ms / TUNE_REP;
        }
    }
    int b = -1;
    for (int c = 0; c < ActTune::NC; ++c) {
        T.t_us[c] = best_us[c];
        if (best_us[c] < 1e29f && (b < 0 || best_us[c] < best_us[b])) b = c;
    }
    // run-off: candidates within 3 % of the winner meet it again, alternating, over longer runs (the first pass times
    // 4 launches per sample; a pick that is wrong by noise costs a whole run 2 - 4 %)
    if (b >= 0) {
        constexpr int RUN_REP = 16;
        int second = -1;
        for (int c = 0; c < ActTune::NC; ++c)
            if (c != b && best_us[c] < 1.03f * best_us[b] && (second < 0 || best_us[c] < best_us[second])) second = c;
        if (second >= 0) {
            float ro[2] = {1e30f, 1e30f};
            const int pair[2] = {b, second};
            for (int round = 0; round < 3; ++round)
                for (int q = 0; q < 2; ++q) {
                    launch_act_as(cand_geo[pair[q]], t, st);
                    (void)hipEventRecord(e0, st);
                    for (int r = 0; r < RUN_REP; ++r) launch_act_as(cand_geo[pair[q]], t, st);
                    (void)hipEventRecord(e1, st);
                    if (hipEventSynchronize(e1) != hipSuccess) { (void)hipGetLastError(); continue; }
                    float ms = 0.f;
                    if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && 1e3f * ms / RUN_REP < ro[q]) ro[q] = 1e3f * ms / RUN_REP;
                }
            if (ro[1] < ro[0]) b = second;
        }
    }
    if (b >= 0) T.best = cand_geo[b];
    // second dimension: the XCD grid of the block -> tile map, with the chosen geometry (the traffic model's choice is
    // one of the four; which one is fastest also depends on how the panels fall onto the memory channels)
    float xi_us[5] = {1e30f, 1e30f, 1e30f, 1e30f, 1e30f};
    static const int cand_xi[5] = {8, 4, 2, 1, -1};
    static const bool tune_xi = !(bm::dbg("xcd_map") || (bm::dbg("tune_xcd") && atoi(bm::dbg("tune_xcd")) == 0));
    T.xi = -1;                                   // the slab order unless a grid is measurably (>= 2 %) faster
    if (tune_xi) {
        for (int round = 0; round < TUNE_ROUNDS; ++round)
            for (int c = 0; c < 5; ++c) {
                t.map_xi = cand_xi[c];
                launch_act_as(T.best, t, st);
                (void)hipEventRecord(e0, st);
                for (int r = 0; r < TUNE_REP; ++r) launch_act_as(T.best, t, st);
                (void)hipEventRecord(e1, st);
                if (hipEventSynchronize(e1) != hipSuccess) { (void)hipGetLastError(); continue; }
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && 1e3f * ms / TUNE_REP < xi_us[c]) xi_us[c] = 1e3f * ms / TUNE_REP;
            }
        int bx = 0;
        for (int c = 1; c < 4; ++c) if (xi_us[c] < xi_us[bx]) bx = c;
        if (xi_us[bx] < 0.98f * xi_us[4]) T.xi = cand_xi[bx];
    } else T.xi = 0;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    static const bool log = bm::dbg("tune_log") != nullptr;
    if (log)
        fprintf(stderr, "bm355 tune: act I=%d J=%d K=%d+%d flags=%lld -> geometry %d (us, dma: 8w %.1f, 64x32 %.1f, 32x32 %.1f, 32x32/bk32 %.1f; "
                        "reg: 8w %.1f, 64x32 %.1f, 32x32 %.1f, 32x32/bk32 %.1f; 64x64 8w: %.1f; 64x32/bk32 x3: %.1f, x2: %.1f; 64x64/bk32 x2: %.1f)\n",
                a.I, a.J, a.K1, a.K2, flags, T.best, T.t_us[0] > 1e29f ? -1.f : T.t_us[0], T.t_us[1] > 1e29f ? -1.f : T.t_us[1], T.t_us[2], T.t_us[3],
                T.t_us[4], T.t_us[5] > 1e29f ? -1.f : T.t_us[5], T.t_us[6], T.t_us[7], T.t_us[8] > 1e29f ? -1.f : T.t_us[8],
                T.t_us[9] > 1e29f ? -1.f : T.t_us[9], T.t_us[10] > 1e29f ? -1.f : T.t_us[10], T.t_us[11] > 1e29f ? -1.f : T.t_us[11]);
    if (log && tune_xi)
        fprintf(stderr, "bm355 tune: act I=%d J=%d K=%d+%d flags=%lld -> tile map %d (-1 slab, else XCD grid xi; us: slab %.1f, 8x1 %.1f, 4x2 %.1f, 2x4 %.1f, 1x8 %.1f)\n",
                a.I, a.J, a.K1, a.K2, flags, T.xi, xi_us[4], xi_us[0], xi_us[1], xi_us[2], xi_us[3]);
}
// fast-binary launches: three tile geometries (2: 64 x 32 tiles, two workgroups per CU; 4: 64 x 64, one; 8: 128 x 32
// with 8 waves), measured once per shape like the fp32 ones.  Which one wins follows the tile count and K, not one
// rule: 20000 AIS chains take 2, the 3072 x 256 x 5000 top-down pass of BASELINE configs[2] takes 8 or 2 (68 / 71 us)
// where 4 needs 117 us (192 tiles on 256 CUs).  BM355_DEBUG=bf3_geo=8|4|2 forces one.
static inline void launch_act_bf3(const ActArgs &a, hipStream_t st) {
    static int geo_env = -1;
    if (geo_env < 0) { const char *e = bm::dbg("bf3_geo"); geo_env = e ? atoi(e) : 0; }
    if (geo_env) { launch_act_bf3_as(geo_env, a, st); return; }
    static std::mutex mu;
    static std::map<std::array<long long, 6>, int> table;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const long long flags = (long long)((a.sample ? 1 : 0) | (a.kind << 1) | (a.rowacc ? 32 : 0) | (a.rowdot_out ? 256 : 0) |
                                        (a.dot_mat ? 512 : 0) | (a.states ? 1024 : 0) | (a.means ? 2048 : 0));
    const std::array<long long, 6> key = {a.I, a.J, a.b3.K1, a.b3.K2, flags, (long long)dev};
    std::lock_guard<std::mutex> lk(mu);
    int &geo = table[key];
    if (!geo) {
        static const int cand[3] = {2, 8, 4};
        constexpr int TUNE_REP = 4, TUNE_ROUNDS = 3;
        geo = tile_grid<GeoBf3S>(a.I, a.J) >= 1024 ? 2 : 8;
        ActArgs t;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (tune_redirect(a, t) && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
            float best[3] = {1e30f, 1e30f, 1e30f};
            for (int round = 0; round < TUNE_ROUNDS; ++round)
                for (int c = 0; c < 3; ++c) {
                    launch_act_bf3_as(cand[c], t, st);
                    (void)hipEventRecord(e0, st);
                    for (int r = 0; r < TUNE_REP; ++r) launch_act_bf3_as(cand[c], t, st);
                    (void)hipEventRecord(e1, st);
                    if (hipEventSynchronize(e1) != hipSuccess) { (void)hipGetLastError(); continue; }
                    float ms = 0.f;
                    if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && 1e3f * ms / TUNE_REP < best[c]) best[c] = 1e3f * ms / TUNE_REP;
                }
            int b = 0;
            for (int c = 1; c < 3; ++c) if (best[c] < best[b]) b = c;
            if (best[b] < 1e29f) geo = cand[b];
            static const bool log = bm::dbg("tune_log") != nullptr;
            if (log) fprintf(stderr, "bm355 tune: bf16x3 act I=%d J=%d K=%d+%d flags=%lld -> geometry %d (us: 64x32 %.1f, 128x32 8w %.1f, 64x64 %.1f)\n",
                             a.I, a.J, a.b3.K1, a.b3.K2, flags, geo, best[0], best[1], best[2]);
        } else (void)hipGetLastError();
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
    launch_act_bf3_as(geo, a, st);
}
static inline void launch_act_f32(const ActArgs &a, hipStream_t st);
static inline void launch_act(const ActArgs &a, hipStream_t st) {
    if (a.b3.K1 > 0) { launch_act_bf3(a, st); return; }
    if (a.fe_flip) { launch_act_fe(a, st); return; }     // the h0 pass of a fused metric fetch: its own kernel flavour
    if (a.prev || a.maxdiff || a.skip || a.chk_ctl || a.acc_init) { launch_act_mf(a, st); return; }   // a mean-field pass: its own flavour
    if (a.lit && a.kind == 0) { launch_act_lit(a, st); return; }   // literal tf.sigmoid: its own kernel flavour
    launch_act_f32(a, st);
    // fast-binary mode, an fp32 launch whose sampled states the NEXT launches read as a bf16 shadow: converted here (the
    // strip kernel writes its shadow itself; keeping the branch out of the fp32 epilogue is worth ~0.2 us per launch)
    if (a.states16 && a.states)
        hipLaunchKernelGGL(shadow16_kernel, dim3(256), dim3(256), 0, st, (const float *)a.states, a.ldo, a.J, a.I, a.states16, a.ld16);
}
static inline void launch_act_f32(const ActArgs &a, hipStream_t st) {
    const int ov = act_geo_override() ? act_geo_override() : a.geo_hint;
    if (ov) { launch_act_as(ov, a, st); return; }
    static std::mutex mu;
    static std::map<std::array<long long, 6>, ActTune> table;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const long long flags = (long long)((a.sample ? 1 : 0) | (a.kind << 1) | (a.prev ? 16 : 0) | (a.rowacc ? 32 : 0) |
                                        (a.acc_init ? 64 : 0) | (a.p_xm ? 128 : 0) | (a.rowdot_out ? 256 : 0) |
                                        (a.dot_mat ? 512 : 0) | (a.negmeans ? 1024 : 0));
    const std::array<long long, 6> key = {a.I, a.J, a.K1, a.K2, flags, (long long)dev};
    int geo, xi;
    {
        std::lock_guard<std::mutex> lk(mu);
        ActTune &T = table[key];
        if (!T.best) tune_act_shape(a, st, T, flags);
        geo = T.best; xi = T.xi;
    }
    if (xi && !a.map_xi) {
        ActArgs a2 = a;
        a2.map_xi = xi;
        launch_act_as(geo, a2, st);
        return;
    }
    launch_act_as(geo, a, st);
}

// ---- grad_kernel geometry choice: 4 waves of 32 x 32 or 8 waves of 32 x 16 (bit-identical results), measured
// once per shape like the act geometries (tune_act_shape), on scratch copies of every buffer the kernel writes.
// BM355_DEBUG=grad_geo=4|8 forces one.
template <class G, int STG, int MINB = 1>
static inline void launch_grad_geo(const GradArgs &g_in, hipStream_t st) {
    const GradArgs &g = g_in;
    TileMap tmap;
    {
        const double kt = (double)g.Kpos + (double)g.Kneg;
        tmap = make_tile_map((g.I + G::TI - 1) / G::TI, (g.J + G::TJ - 1) / G::TJ, kt * G::TI * 4.0, kt * G::TJ * 4.0, g.map_xi);
    }
    const bool fast = operand_fast(g.Ppos, KM, g.Kpos) && operand_fast(g.Qpos, KM, g.Kpos) &&
                      operand_fast(g.Pneg, KM, g.Kneg) && operand_fast(g.Qneg, KM, g.Kneg);
    const dim3 grid(tile_grid<G>(g.I, g.J) + g.nbias), blk(G::NT);
    if (fast) hipLaunchKernelGGL((grad_kernel<G, true, 0, STG, MINB>), grid, blk, 0, st, g, tmap);
    else      hipLaunchKernelGGL((grad_kernel<G, false, 0, STG_DMA, MINB>), grid, blk, 0, st, g, tmap);
}
// geo: 4 | 8 waves, + 100 for register staging of the full chunks
static inline void launch_grad_as(int geo, const GradArgs &g, hipStream_t st) {
    // 9: 8 waves, BK = 32, two workgroups per CU - for outputs of many tiles per CU and a short K (3072 x 5000 x 512:
    // a tile's fill and read-modify-write epilogue take as long as its K loop)
    if (geo == 9 && g.nbias == 0) launch_grad_geo<GeoGrad8h, STG_DMA, 4>(g, st);       // 4 waves per SIMD = 2 workgroups per CU
    else if (geo == 9)   launch_grad_geo<GeoGrad8, STG_DMA>(g, st);
    else if (geo == 208) launch_grad_geo<GeoGrad8, STG_DMAH>(g, st);
    else if (geo == 108) launch_grad_geo<GeoGrad8, STG_REG>(g, st);
    else if (geo == 104) launch_grad_geo<GeoGrad, STG_REG>(g, st);
    else if (geo == 8)   launch_grad_geo<GeoGrad8, STG_DMA>(g, st);
    else                 launch_grad_geo<GeoGrad, STG_DMA>(g, st);
}
static inline int tune_grad_shape(const GradArgs &g, hipStream_t st) {
    // the tile workgroups only: scratch W / dW / raw (the update is not idempotent), no bias groups
    static TuneScratch pool;
    const size_t mat = (((size_t)g.J * (size_t)g.ldw) + 3) & ~(size_t)3;
    float *s = pool.get(4 * mat);
    if (!s) return 4;
    GradArgs t = g;
    t.nbias = 0; t.pen = nullptr; t.Wt = nullptr;
    t.W = s; t.dW = s + mat; t.raw = s + 2 * mat; t.raw2 = s + 3 * mat;
    (void)hipMemsetAsync(s, 0, 4 * mat * sizeof(float), st);
#ifdef BM_PROBE
    t.dbg = nullptr;
#endif
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); return 4; }
    constexpr int NC = 5;
    const int cand[NC] = {4, 8, 104, 108, 9};          // 208 (half-wave DMA) is forceable, never the fastest
    float best_us[NC] = {1e30f, 1e30f, 1e30f, 1e30f, 1e30f};
    for (int round = 0; round < 3; ++round)
        for (int c = 0; c < NC; ++c) {
            launch_grad_as(cand[c], t, st);
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < 4; ++r) launch_grad_as(cand[c], t, st);
            (void)hipEventRecord(e1, st);
            if (hipEventSynchronize(e1) != hipSuccess) { (void)hipGetLastError(); continue; }
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && 250.f * ms < best_us[c]) best_us[c] = 250.f * ms;
        }
    int b = 0;
    for (int c = 1; c < NC; ++c) if (best_us[c] < best_us[b]) b = c;
    const int best = cand[b];
    // the XCD grid of the block -> tile map with that geometry (see tune_act_shape)
    float xi_us[5] = {1e30f, 1e30f, 1e30f, 1e30f, 1e30f};
    static const int cand_xi[5] = {8, 4, 2, 1, -1};
    static const bool tune_xi = !(bm::dbg("xcd_map") || (bm::dbg("tune_xcd") && atoi(bm::dbg("tune_xcd")) == 0));
    int xi = 9;                                  // 9 = the slab order (map_xi -1), unless a grid is >= 2 % faster
    if (tune_xi) {
        for (int round = 0; round < 3; ++round)
            for (int c = 0; c < 5; ++c) {
                t.map_xi = cand_xi[c];
                launch_grad_as(best, t, st);
                (void)hipEventRecord(e0, st);
                for (int r = 0; r < 4; ++r) launch_grad_as(best, t, st);
                (void)hipEventRecord(e1, st);
                if (hipEventSynchronize(e1) != hipSuccess) { (void)hipGetLastError(); continue; }
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && 250.f * ms < xi_us[c]) xi_us[c] = 250.f * ms;
            }
        int bx = 0;
        for (int c = 1; c < 4; ++c) if (xi_us[c] < xi_us[bx]) bx = c;
        if (xi_us[bx] < 0.98f * xi_us[4]) xi = cand_xi[bx];
    } else xi = 0;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    static const bool log = bm::dbg("tune_log") != nullptr;
    if (log)
        fprintf(stderr, "bm355 tune: grad I=%d J=%d K=%d+%d form=%d fused=%d -> geometry %d (us, dma: 4w %.1f, 8w %.1f; reg: 4w %.1f, 8w %.1f; 8w bk32 x2: %.1f), "
                        "tile map %d (9 slab, else XCD grid xi; us: slab %.1f, 8x1 %.1f, 4x2 %.1f, 2x4 %.1f, 1x8 %.1f)\n",
                g.I, g.J, g.Kpos, g.Kneg, g.form, g.fused, best, best_us[0], best_us[1], best_us[2], best_us[3], best_us[4], xi, xi_us[4], xi_us[0], xi_us[1], xi_us[2], xi_us[3]);
    return best + 1000 * xi;
}
static inline void launch_grad(const GradArgs &g_in, hipStream_t st) {
    static int fetch_env = -1, geo_env = -1;          // BM355_DEBUG=grad_fetch=0|1, BM355_DEBUG=grad_geo=4|8 override (experiments)
    if (fetch_env < 0) { const char *e = bm::dbg("grad_fetch"); fetch_env = e ? 2 + atoi(e) : 0; }
    if (geo_env < 0) { const char *e = bm::dbg("grad_geo"); geo_env = e ? atoi(e) : 0; }
    GradArgs g = g_in;
    g.fetch_at_fill = fetch_env >= 2 ? fetch_env - 2 : 0;      // measured (same box, 784x1024x512): epilogue 66.6 us/update, fill 67.5
    int geo = geo_env;
    if (!geo) {
        static std::mutex mu;
        static std::map<std::array<long long, 7>, int> table;
        int dev = 0;
        (void)hipGetDevice(&dev);
        const std::array<long long, 7> key = {g.I, g.J, g.Kpos, g.Kneg, (long long)(g.form | (g.fused << 1)), (long long)g.ldw, (long long)dev};
        std::lock_guard<std::mutex> lk(mu);
        int &b = table[key];
        if (!b) b = tune_grad_shape(g, st);
        geo = b;
    }
    if (geo >= 1000) { if (!g.map_xi) g.map_xi = (geo / 1000 == 9) ? -1 : geo / 1000; geo %= 1000; }
    launch_grad_as(geo, g, st);
}

static inline void launch_fe_hidden(const FeArgs &f, hipStream_t st) {
    const bool fast = operand_fast(f.P, KM, f.K) && operand_fast(f.Q, XM, f.K);
    const dim3 grid(tile_grid<GeoAct>(f.I, f.J)), blk(NT);
    if (fast) hipLaunchKernelGGL((fe_hidden_kernel<true>), grid, blk, 0, st, f);
    else      hipLaunchKernelGGL((fe_hidden_kernel<false>), grid, blk, 0, st, f);
}

}  // namespace bm
