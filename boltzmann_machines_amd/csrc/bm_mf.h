// bm_mf.h — the mean-field loop of a two-layer DBM (dbm.py:429-478) as ONE persistent kernel.
//
// The launch-per-layer form (bm_dbm.hip mean_field) runs 2 kernels per sweep; at the BASELINE configs[3] shape
// (512 rows, 784-512-1024) each holds 3.4 us of matrix work under ~8 us of per-launch fixed cost (launch gap,
// prologue, first-chunk latency after the kernel-boundary L2 invalidate, epilogue): 92 dependent launches = 1.1 of
// the 1.42 ms of a DBM update, 0.28 of the fp32-MFMA roof.  Mean-field rows are independent, and both weight
// operands of a sweep are the same 2 MB matrix W1 - so here
//   * XCD x owns rows [x RB, (x+1) RB) of the minibatch (RB = N / 8); its 32 workgroups (one per CU, recognised by
//     HW_REG_XCC_ID, column slice by a per-XCD ticket) split the COLUMNS: in the h1 update workgroup c computes columns
//     [16 c, 16 c + 16) of mu1, in the h2 update columns [32 c, 32 c + 32) of mu2;
//   * the two weight slices a workgroup needs - W1^T[:, 16 columns] (1024 x 16) and W1[:, 32 columns] (512 x 32) -
//     are loaded into LDS ONCE and stay there for all sweeps (152 KiB with the bank padding);
//   * the activations stream: a lane's MFMA B operand of one 16-k block is 16 contiguous bytes of ITS row of mu, read
//     straight from global memory into registers (L1-bypassing loads, 8 blocks ahead), no LDS staging: rows are
//     private to a wave, there is nothing to share;
//   * mu1 -> mu2 -> mu1 are handed over inside the XCD through its L2: plain stores, `s_waitcnt vmcnt(0)`, a workgroup
//     barrier, ONE relaxed agent-scope atomic add on the XCD's arrive counter; the consumers poll that counter
//     (one lane, sc1 loads) and read the data with sc1 loads.  No cache flush or invalidate is needed because producer
//     and consumer share the L2 by construction; no grid barrier, no kernel boundary;
//   * the loop condition max|mu_new - mu| > tol is over ALL rows: every XCD's last arriver publishes its residual with
//     device-scope atomics, and every workgroup reads the decision for sweep s-2 under the matrix work of sweep s
//     (speculation depth 1: sweep s-1 may turn out to be one too many - it wrote the other buffer and is discarded),
//     so the executed trip count is the reference's;
//   * arithmetic is the canonical chain of bm_gemm.h (acc = X.W0 hoisted, then k = 16 m + 4 g + j ascending blocks;
//     the same sigmoid): results are bit-identical to the launch-per-layer path.
// MEASURED (MI355X, tools/bench_mf.py with BM355_MF_DEBUG=1, 512 rows): bit-identical to the per-layer path in every
// test, but NOT faster - per sweep 10.4 us in the h1 update's K loop (4.3 us of matrix work), 8.2 us in the h2 update's
// (3.4 us), 1.5 - 2.5 us per hand-over wait, 1.7 us to publish = ~24 us, what the two launches take.  The K loops are
// bound by the L2 -> CU delivery of the activation rows: all 32 CUs of an XCD stream the SAME 64 rows in lockstep and
// get ~25 GB/s each; sc1 or L1-cached loads, 8 or 16 blocks in flight, loads pinned between the MFMAs and LDS
// fragments read a block ahead all measured the same (10.4 +- 0.4 us).  It also takes every CU (152 KiB of LDS), so
// the PCD sweeps of the same update can no longer run next to the mean-field.  Hence OPT-IN (bm_dbm_set_mf_persistent),
// kept as the measured form of "fuse the sweeps": what it would need is activations that do not cross the L2 per CU
// (row blocks per CU = weights streamed instead, which is the per-layer kernel again).
// Correctness does not depend on where the hardware places workgroups: a workgroup that finds its XCD over-subscribed
// (ticket >= 32), or any bounded wait that expires, raises `status` and every workgroup leaves; the host then runs the
// launch-per-layer path from the untouched persistent mu (the kernel only writes work buffers).
#pragma once
#include "bm_gemm.h"
#include "bm_numerics.h"

namespace bm {

constexpr int MFP_H1 = 512, MFP_H2 = 1024;       // the shape family this kernel tiles exactly: 32 x 16 and 32 x 32 columns
constexpr int MFP_P1 = 20, MFP_P2 = 36;          // LDS row pitches (floats) of the two weight slices: rows k and k + 4
                                                 // (lane groups g, g + 1 of one ds_read_b32) land on different bank halves
constexpr int MFP_D = 8;                         // k blocks of the activation stream in flight per lane
constexpr int MFP_MAXS = 128;                    // sweeps the synchronisation block has room for

struct MfSync {                                  // zeroed before every launch
    unsigned ticket[8];
    unsigned bar[8][2];                          // monotone arrive counters per XCD: after the h1 / h2 update
    unsigned xres[8][MFP_MAXS];                  // per XCD and sweep: max residual (float bits; residuals are >= 0)
    unsigned gres[MFP_MAXS], gcnt[MFP_MAXS];     // per sweep: max over the XCDs, number of XCDs that contributed
    int status;                                  // != 0: abandoned (1 timeout, 2 over-subscribed XCD)
};

struct MfpArgs {
    int N, RB;                                   // rows, rows per XCD
    const float *xw0; int ld_x;                  // [N][H1] hoisted X.W0
    const float *Wt1; int ld_wt;                 // [H2][H1]: k = h2, i = h1
    const float *W1; int ld_w;                   // [H1][H2]: k = h1, i = h2
    const float *hb0, *hb1;
    float *mu1[3], *mu2[3]; int ld1, ld2;        // [0]: the persistent mu (read only), [1], [2]: work buffers
    MfCtl *ctl; float tol; int max_steps;
    MfSync *sy;
    long long timeout;                           // wall_clock64 ticks
    long long *dbg;                              // optional [MFP_MAXS][8] wall-clock stamps of workgroup (XCD 0, slice 0): tools/bench_mf.py
};

__device__ __forceinline__ unsigned mfp_ld(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one lane waits until *ctr >= target (wrap-safe); false on abort / timeout (status raised)
__device__ __forceinline__ bool mfp_wait(const unsigned *ctr, unsigned target, const MfpArgs &a) {
    const long long t0 = wall_clock64();
    int spins = 0;
    while ((int)(mfp_ld(ctr) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63) == 0) {
            if (__hip_atomic_load(&a.sy->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (wall_clock64() - t0 > a.timeout) { __hip_atomic_store(&a.sy->status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        }
    }
    return true;
}

__global__ __launch_bounds__(256, 1) void mf_persistent_kernel(MfpArgs a) {
    __shared__ __attribute__((aligned(16))) float sP1[MFP_H2 * MFP_P1];      // W1^T[k = h2][16 columns of h1]
    __shared__ __attribute__((aligned(16))) float sP2[MFP_H1 * MFP_P2];      // W1[k = h1][32 columns of h2]
    __shared__ float s_red[4];
    __shared__ int s_i[4];                       // [0] ticket / final steps, [1] h2-barrier, [2] decision, [3] h1-barrier
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int x = (int)(__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 7u);          // HW_REG_XCC_ID
    if (tid == 0) s_i[0] = (int)__hip_atomic_fetch_add(&a.sy->ticket[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int c = s_i[0];
    if (c >= 32) {                               // more than 32 workgroups on this XCD: not the placement this kernel needs
        if (tid == 0) __hip_atomic_store(&a.sy->status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // ---- the resident weight slices
    for (int e = tid; e < MFP_H2 * 4; e += 256) {             // 16 columns = 4 float4 per row
        const int k = e >> 2, c4 = e & 3;
        *reinterpret_cast<float4 *>(sP1 + k * MFP_P1 + 4 * c4) =
            *reinterpret_cast<const float4 *>(a.Wt1 + (size_t)k * a.ld_wt + 16 * c + 4 * c4);
    }
    for (int e = tid; e < MFP_H1 * 8; e += 256) {             // 32 columns = 8 float4 per row
        const int k = e >> 3, c4 = e & 7;
        *reinterpret_cast<float4 *>(sP2 + k * MFP_P2 + 4 * c4) =
            *reinterpret_cast<const float4 *>(a.W1 + (size_t)k * a.ld_w + 32 * c + 4 * c4);
    }
    __syncthreads();
    const bool active = 16 * w < a.RB;           // wave w owns rows 16 w .. 16 w + 15 of the XCD's block
    const int j = x * a.RB + 16 * w + l15;       // this lane's row (MFMA column l15)
    const int i1 = 16 * c + 4 * g;               // its 4 consecutive outputs of the h1 update
    const int i2 = 32 * c + 4 * g;               // ... of the h2 update: i2 + 16 t + r
    f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, b1 = x0, b2[2] = {x0, x0};
    if (active) {
        x0 = *reinterpret_cast<const f32x4 *>(a.xw0 + (size_t)j * a.ld_x + i1);
        b1 = *reinterpret_cast<const f32x4 *>(a.hb0 + i1);
        b2[0] = *reinterpret_cast<const f32x4 *>(a.hb1 + i2);
        b2[1] = *reinterpret_cast<const f32x4 *>(a.hb1 + i2 + 16);
    }
    const unsigned row1 = (unsigned)(((size_t)j * a.ld1) * 4), row2 = (unsigned)(((size_t)j * a.ld2) * 4);   // byte offsets of row j
    const size_t bytes1 = (size_t)a.N * a.ld1 * 4, bytes2 = (size_t)a.N * a.ld2 * 4;
    auto rs = [](const float *p, size_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, (int)bytes, 0x00020000); };
    // The activation stream: another CU of this XCD wrote the data, through the shared L2.  A lane's 16 bytes are half
    // of a 64-byte segment and two consecutive k blocks share a 128-byte line: as L1-bypassing (sc1) loads every line
    // is requested from the L2 twice, half-filled - measured 9.7 us per h1 update for 4.3 us of matrix work.  So the
    // loads go THROUGH the L1 (the second block of a line hits it) and the L1 is invalidated once per phase, after
    // the hand-over wait (`buffer_inv sc1`, one wave + workgroup barrier).  MFP_SC1=1 restores the bypassing loads.
#ifndef MFP_SC1
#define MFP_SC1 0
#endif
    auto ld16 = [](__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, MFP_SC1 ? 16 : 0));
    };
    auto l1_invalidate = [&]() {
        if (!MFP_SC1) {
            if (w == 0) asm volatile("buffer_inv sc1" ::: "memory");
            __syncthreads();
        }
    };
    int steps = 0, done = a.ctl->done;           // the step-0 condition was evaluated by the caller
    int final_steps = -1;
    bool ok = true;
    if (!done) {
        for (int s = 0; s < a.max_steps && s < MFP_MAXS; ++s) {
            const int src = s == 0 ? 0 : 1 + ((s - 1) & 1), dst = 1 + (s & 1);
#define MFP_STAMP(n) do { if (a.dbg && x == 0 && c == 0 && tid == 0) a.dbg[s * 8 + (n)] = wall_clock64(); } while (0)
            MFP_STAMP(0);
            // h2 update of sweep s-1 complete on this XCD
            if (s > 0) {
                if (tid == 0) s_i[1] = mfp_wait(&a.sy->bar[x][1], 32u * (unsigned)s, a) ? 1 : 0;
                __syncthreads();
                if (!s_i[1]) { ok = false; break; }
                l1_invalidate();
            }
            MFP_STAMP(1);
            // ---------------- h1 update: mu1[dst] = sigmoid(X.W0 + mu2[src].W1^T + hb0)
            f32x4 acc = x0, old1 = {0.f, 0.f, 0.f, 0.f};
            if (active) {
                const __amdgpu_buffer_rsrc_t rq = rs(a.mu2[src], bytes2), ro = rs(a.mu1[src], bytes1);
                old1 = ld16(ro, row1 + (unsigned)i1 * 4u, 0);
                f32x4 q[MFP_D];
#pragma unroll
                for (int d = 0; d < MFP_D; ++d) q[d] = ld16(rq, row2 + 16u * (unsigned)g, 64u * (unsigned)d);
                // the weight fragments of block m+1 are read from LDS while the MFMAs of block m run
                float pc[4], pn[4];
                {
                    const float *p = sP1 + (4 * g) * MFP_P1 + l15;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) pc[jj] = p[jj * MFP_P1];
                }
#pragma unroll 1
                for (int m0 = 0; m0 < MFP_H2 / 16; m0 += MFP_D) {
#pragma unroll
                    for (int d = 0; d < MFP_D; ++d) {
                        const int m = m0 + d;
                        const f32x4 cur = q[d];
                        // (unconditional, clamped: a load under a branch makes hipcc wait for ALL loads in flight at every
                        //  use - it cannot count a load that may not have been issued - which serialised the stream)
                        q[d] = ld16(rq, row2 + 16u * (unsigned)g, 64u * (unsigned)min(m + MFP_D, MFP_H2 / 16 - 1));
                        const float *p = sP1 + (16 * min(m + 1, MFP_H2 / 16 - 1) + 4 * g) * MFP_P1 + l15;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) pn[jj] = p[jj * MFP_P1];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pc[jj], cur[jj], acc, 0, 0, 0);
                        // keep the block's loads where they are written: hipcc otherwise gathers the eight global loads at
                        // the end of the unrolled body and waits for all of them there
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) pc[jj] = pn[jj];
                    }
                }
            }
            MFP_STAMP(2);
            // the decision for sweep s-2, read under the matrix work above: converged there -> sweep s-1 was one too
            // many (it wrote the other buffer) and this one must not overwrite the result
            if (s >= 2) {
                if (tid == 0) {
                    int v = mfp_wait(&a.sy->gcnt[s - 2], 8u, a) ? 1 : 0;
                    if (v && !(__uint_as_float(mfp_ld(&a.sy->gres[s - 2])) > a.tol)) v = 2;
                    s_i[2] = v;
                }
                __syncthreads();
                const int v = s_i[2];
                if (v == 0) { ok = false; break; }
                if (v == 2) { final_steps = s - 1; break; }
            }
            MFP_STAMP(3);
            float dmax = 0.f;
            if (active) {
                f32x4 mnew;
#pragma unroll
                for (int r = 0; r < 4; ++r) { mnew[r] = sigmoid(acc[r] + b1[r]); dmax = fmaxf(dmax, fabsf(mnew[r] - old1[r])); }
                *reinterpret_cast<f32x4 *>(a.mu1[dst] + (size_t)j * a.ld1 + i1) = mnew;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                (void)__hip_atomic_fetch_add(&a.sy->bar[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_i[3] = mfp_wait(&a.sy->bar[x][0], 32u * (unsigned)(s + 1), a) ? 1 : 0;
            }
            __syncthreads();
            if (!s_i[3]) { ok = false; break; }
            l1_invalidate();
            MFP_STAMP(4);
            // ---------------- h2 update: mu2[dst] = sigmoid(mu1[dst].W1 + hb1)
            if (active) {
                f32x4 ac2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                const __amdgpu_buffer_rsrc_t rq = rs(a.mu1[dst], bytes1), ro = rs(a.mu2[src], bytes2);
                const f32x4 o0 = ld16(ro, row2 + (unsigned)i2 * 4u, 0), o1 = ld16(ro, row2 + (unsigned)(i2 + 16) * 4u, 0);
                f32x4 q[MFP_D];
#pragma unroll
                for (int d = 0; d < MFP_D; ++d) q[d] = ld16(rq, row1 + 16u * (unsigned)g, 64u * (unsigned)d);
                float pc[8], pn[8];
                {
                    const float *p = sP2 + (4 * g) * MFP_P2 + l15;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) { pc[2 * jj] = p[jj * MFP_P2]; pc[2 * jj + 1] = p[jj * MFP_P2 + 16]; }
                }
#pragma unroll 1
                for (int m0 = 0; m0 < MFP_H1 / 16; m0 += MFP_D) {
#pragma unroll
                    for (int d = 0; d < MFP_D; ++d) {
                        const int m = m0 + d;
                        const f32x4 cur = q[d];
                        q[d] = ld16(rq, row1 + 16u * (unsigned)g, 64u * (unsigned)min(m + MFP_D, MFP_H1 / 16 - 1));
                        const float *p = sP2 + (16 * min(m + 1, MFP_H1 / 16 - 1) + 4 * g) * MFP_P2 + l15;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) { pn[2 * jj] = p[jj * MFP_P2]; pn[2 * jj + 1] = p[jj * MFP_P2 + 16]; }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            ac2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pc[2 * jj], cur[jj], ac2[0], 0, 0, 0);
                            ac2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pc[2 * jj + 1], cur[jj], ac2[1], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) pc[jj] = pn[jj];
                    }
                }
                f32x4 m0v, m1v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m0v[r] = sigmoid(ac2[0][r] + b2[0][r]); dmax = fmaxf(dmax, fabsf(m0v[r] - o0[r]));
                    m1v[r] = sigmoid(ac2[1][r] + b2[1][r]); dmax = fmaxf(dmax, fabsf(m1v[r] - o1[r]));
                }
                *reinterpret_cast<f32x4 *>(a.mu2[dst] + (size_t)j * a.ld2 + i2) = m0v;
                *reinterpret_cast<f32x4 *>(a.mu2[dst] + (size_t)j * a.ld2 + i2 + 16) = m1v;
            }
            MFP_STAMP(5);
            // residual of this workgroup -> the XCD's cell; the XCD's last arriver publishes it for the loop condition
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));
            if (lane == 0) s_red[w] = dmax;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const float m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
                // relaxed atomics only (an agent-scope release would write the XCD's whole L2 back): every atomic is a
                // RETURNING one whose value is consumed before the next is issued, so they are performed in this order
                unsigned o1 = __hip_atomic_fetch_max(&a.sy->xres[x][s], __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("" :: "v"(o1));
                const unsigned prev = __hip_atomic_fetch_add(&a.sy->bar[x][1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (prev + 1u == 32u * (unsigned)(s + 1)) {
                    const unsigned v = mfp_ld(&a.sy->xres[x][s]);
                    unsigned o2 = __hip_atomic_fetch_max(&a.sy->gres[s], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("" :: "v"(o2));
                    (void)__hip_atomic_fetch_add(&a.sy->gcnt[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            MFP_STAMP(6);
            steps = s + 1;
        }
#undef MFP_STAMP
        // the loop ran out of sweeps (or was abandoned): the decisions of the last two sweeps are still open
        if (ok && final_steps < 0) {
            final_steps = steps;
            if (tid == 0) {
                int v = 1;
                for (int s = (steps >= 2 ? steps - 2 : 0); s < steps && v == 1; ++s) {
                    if (!mfp_wait(&a.sy->gcnt[s], 8u, a)) { v = 0; break; }
                    if (!(__uint_as_float(mfp_ld(&a.sy->gres[s])) > a.tol)) { s_i[0] = s + 1; v = 2; }
                }
                s_i[2] = v;
            }
            __syncthreads();
            if (s_i[2] == 0) ok = false;
            if (s_i[2] == 2) final_steps = s_i[0];
        }
    } else {
        final_steps = 0;
    }
    if (ok && x == 0 && c == 0 && tid == 0) {
        // steps / done exactly as mf_ctl_kernel leaves them: done = the last counted sweep's residual did not exceed tol
        int dn = done;
        if (!done) dn = (final_steps >= 1 && !(__uint_as_float(mfp_ld(&a.sy->gres[final_steps - 1])) > a.tol)) ? 1 : 0;
        a.ctl->steps = final_steps;
        a.ctl->done = dn;
    }
}

}  // namespace bm
