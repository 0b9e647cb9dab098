// bm_dbmchain.h — one DBM update's mean-field loop (dbm.py:429-478) AND its persistent-chain sweeps (dbm.py:480-509) as
// workgroups of ONE launch.
//
// Why.  At 784-512-1024 x 512 a mean-field sweep is two propagation passes of 0.54 GFLOP each, 11.5 + 13.1 us as two
// launches of which ~7 us are the two kernel boundaries; 46 - 50 sweeps make 1.2 of the 1.4 ms of an update, and the 15
// passes of the 5 particle sweeps run beside them on a second stream and compete for the same CUs.  Both families are
// ROW LOCAL - a pass reads, of what earlier passes wrote, only the rows it writes itself - which is what bm_chain.h was
// built for: the 8 row blocks (64 rows) of the minibatch and the 8 row blocks of the particles are dealt to the 8 XCD
// teams, passes hand their rows over inside the team's L2, and the kernel boundaries disappear.
//
// What is new against bm_chain.h (plain RBM passes):
//   * two pass families in one launch.  A team's MAIN sequence is the mean-field tiles in dependency order with SLOTS
//     for particle tiles in between - per sweep [h1 tiles][slots][h2 tiles].  The h1 pass has 16 tile columns for the 32
//     workgroups of a team: the other half takes a particle tile each (their own claim counter, their own dependency
//     order), which is about as long as the h1 tiles next to it.  Every tile waits only for tiles EARLIER IN ITS OWN
//     family's order, all of which are claimed by running workgroups: no deadlock for any number >= 1 of resident
//     workgroups per XCD.  When the mean-field loop has ended (or for teams without data rows) workgroups drain the
//     particle counter directly;
//   * passes with a second K segment (h1 of a particle sweep: v.W0 + h2.W1^T), a stored partial pre-activation to start
//     from (the hoisted X.W0 of the mean-field) and the mean-field residual max |mu_new - mu|;
//   * the data-dependent trip count.  The loop ends after the first sweep s whose residual over ALL rows - all teams -
//     is <= tol, i.e. in which NO tile saw an element move by more than tol.  Every tile counts itself, and whether it
//     saw one, with ONE agent-scope atomicAdd on arrived[s] (fire and forget: no wave waits for an atomic).  The word IS the
//     verdict: complete when its low half equals the number of tiles of a sweep, "converged" when its high half is 0.  A
//     tile of sweep s starts when arrived[s - 2] is complete and says go on: it has had a whole sweep to cross the XCDs, nobody waits for it in
//     practice, and at most ONE sweep past the end is executed speculatively.  Sweep s writes buffer s % 3 of three, so
//     the speculative sweep n + 1 overwrites mu_{n-2}, never the result mu_n; tiles of sweep n + 2 find sweep n
//     converged and do not run.  The host learns n from dch_finish_kernel and rotates its buffer handles.
// Results are bit-identical to the per-pass launches (same tile body, same canonical order, same Philox addressing) and
// the executed sweep count is the same number (tests/test_full_size_gpu.py, tests/test_dbm_parity_gpu.py).
#pragma once
#include "bm_chain.h"

namespace bm {

constexpr int DCH_MAXSW = 64;      // mean-field sweeps one launch can hold (a larger max_mf_updates keeps per-pass launches)
constexpr int DCH_MAXPC = 16;      // particle sweeps one launch can hold
constexpr int DCH_MAXTI = 64;      // tile columns per pass (the wait polls one flag per lane)

struct DchPass {
    Operand P1, P2;                // weights of the two K segments (K2 == 0: one segment)
    int K1, K2, p_xm, I, ntile, kind, sample;
    const float *bias, *sigma;
    PhiloxKey key;                 // of sweep 0 (site + 16 t for sweep t)
};

struct DchArgs {
    unsigned gen; int dbg;
    unsigned *flags_mf;            // [8][2 * DCH_MAXSW][DCH_MAXTI]: generation of the launch that completed the tile
    unsigned *flags_pc;            // [8][3 * DCH_MAXPC][DCH_MAXTI]
    unsigned *claim, *claim_zero;  // [8][32]: word 0 main sequence, word 16 particle tiles; claim_zero: a later launch's
    // zeroed per launch: arrived [DCH_MAXSW + 2], per sweep: low 16 bits = tiles of the sweep that have finished, high 16 bits =
    // how many of them saw a residual > tol (ONE atomicAdd per tile, nobody looks at its return value); stop: one word (dch_stopped)
    unsigned *arrived, *stop;
    int *status;
    // mean-field family
    int mf_sweeps; float tol; int arrive_total, J_mf, tiles_mf;
    DchPass mf[2];                 // h1 <- xw0 + mu2.W1^T,  h2 <- mu1.W1
    float *mu[3][2]; int ld_mu[2];
    const float *xw0; int ld_xw0;
    const int *done0;              // the step-0 condition was false: no sweep at all
    // particle family
    int pc_sweeps, J_pc, tiles_pc, slots;
    DchPass pc[3];                 // h1 <- v.W0 + h2.W1^T,  h2 <- h1.W1,  v <- h1.W0^T
    float *pv[2]; int ld_v;
    float *ph[2][2]; int ld_h[2];  // [buffer][layer]
    long long prow0;
    long long *stamps;             // developer timeline (BM355_DEBUG=dch_stamps=file): [block][DCH_STAMP_TILES][8] 100 MHz clock values, else null
};
constexpr int DCH_STAMP_TILES = 160;
static_assert(sizeof(DchArgs) <= 4096, "kernel arguments");

// state operands inside the launch: library-owned matrices (16-byte aligned rows: the host checks it)
__device__ __forceinline__ Operand make_operand_dev(const float *p, int ld, int nx) {
    Operand o;
    o.ptr = p; o.ld = ld; o.nx = nx; o.vec = 1;
    return o;
}
// The stop word counts DOWN so that one memset(0) initialises every control word: 0 = no sweep has converged, otherwise
// 0xffffffff - (first converged sweep), raised with atomicMax.  "a sweep <= lim has converged":
__device__ __forceinline__ bool dch_stopped(unsigned word, unsigned lim) { return word >= 0xffffffffu - lim; }

template <int E, class Rng> struct DchSide : ActSide<E, Rng> {
    static constexpr bool kSplitFill = true, kCohQ = true, kCanAbort = true;
    const unsigned *wfA, *wfB;     // flags of the passes that produced Q1 / Q2 for this row block (null: launch inputs)
    int nA, nB;
    const unsigned *verdict;       // &arrived[sweep - 2], or null (sweeps 1 and 2, particle tiles)
    unsigned *stop;                // mean-field tiles of sweeps >= 3: give up when a sweep <= sweep - 2 has converged
    unsigned lim;                  // sweep - 2
    unsigned total;                // tiles of one sweep over all teams
    unsigned gen;
    int *status;
    unsigned *s_abort;             // one LDS word
    long long *stamp;              // null, or where the end of the wait is recorded
    const float *prev_coh;         // &prev[j][ib0] (written by another workgroup of this launch) or null
    int aborted;
    // One wave polls: lane < nA + nB a producer's flag, lane 62 the verdict, lane 63 the stop word.  Bounded: an expired
    // wait sets the sticky status word and aborts the tile (wrong results, reported by bm_dbm_sync; no hang).
    __device__ __forceinline__ void wait_inputs() {
        aborted = 0;
        if (wfA || wfB || verdict) {                   // workgroup-uniform
            if ((threadIdx.x >> 6) == 0) {
                const int lane = threadIdx.x & 63;
                unsigned spins = 0;
                int ab = 0;
                // The flags live in this XCD's L2 (~0.6 us per poll); the verdict and the stop word come across the
                // fabric (2 - 3 us a load) and would set the pace of EVERY poll: the verdict is normally known before the tile
                // starts (prefetched under the previous tile's epilogue: `verdict` is then null) and otherwise read until
                // it is complete; the stop word every 16th poll (it only matters when the producers of this tile will
                // never run).
                bool vknown = verdict == nullptr;
                for (;;) {
                    unsigned v = gen;
                    if (lane < nA) v = __hip_atomic_load(wfA + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else if (lane < nA + nB) v = __hip_atomic_load(wfB + (lane - nA), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bool ok = v == gen, bad = false;
                    if (lane == 62 && !vknown) {
                        const unsigned vd = __hip_atomic_load(verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        vknown = (vd & 0xffffu) == total; ok = vknown; bad = vknown && (vd >> 16) == 0u;
                        // sweep lim has converged: tell the workgroups whose producers will never run
                        if (bad) (void)__hip_atomic_fetch_max(stop, 0xffffffffu - lim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (lane == 63 && stop && (spins & 15u) == 15u)
                        bad = dch_stopped(__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), lim);
                    if (__any(bad)) { ab = 1; break; }
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 1023u) == 0) {
                        const int st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (st || spins > (1u << 21)) {
                            if (lane == 0 && !st) atomicExch(status, CHAIN_ERR_TIMEOUT);
                            ab = 1;
                            break;
                        }
                    }
                }
                if (lane == 0) *s_abort = (unsigned)ab;
            }
            wg_barrier();
            aborted = __builtin_amdgcn_readfirstlane((int)*s_abort);
        }
        if (stamp && threadIdx.x == 0) *stamp = wall_clock64();
        if (prev_coh && !aborted) {
#pragma unroll
            for (int e = 0; e < E; ++e)
                this->pv[e] = (this->ib0 + e < this->I) ? __hip_atomic_load(prev_coh + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        }
    }
};

// (never hand a REFERENCE to the kernel-argument struct to a helper: every later read of it becomes a flat load, which the
//  compiler must treat as divergent - the pass templates then land in VGPRs and the DMA instructions have no scalar base)
struct DchWait {
    const unsigned *wfA, *wfB; int nA, nB;
    const unsigned *verdict; unsigned *stop; unsigned lim, total;
};

// one output tile of one pass.  Returns through `dmax` the tile's mean-field residual (workgroup-uniform after the caller's
// barrier), through `aborted` whether the tile gave up before touching memory.
template <class G, bool SEG2, int PL>
__device__ __forceinline__ void dch_tile(const ActArgs &a, int i0, int j0, float *smem, const DchWait &wt, unsigned gen, int *status,
                                         unsigned *s_abort, unsigned *claim, unsigned &nxt, const unsigned *stop, unsigned &stop_seen,
                                         const unsigned *pre_ptr, unsigned &pre0, unsigned &pre1, float &dmax, int &aborted,
                                         long long *stamps) {
    constexpr int E = G::E;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w % G::WI, wj = w / G::WI;
    const int g = lane >> 4, l15 = lane & 15;
    const int ib0 = i0 + wi * (16 * G::MI) + g * E;
    const int j = j0 + wj * 16 + l15;
    KRange kr;
    kr.P1 = a.P1; kr.Q1 = a.Q1; kr.K1 = a.K1;
    kr.P2 = a.P2; kr.Q2 = a.Q2; kr.K2 = SEG2 ? a.K2 : 0;
    DchSide<E, typename PhiloxFor<G::MI>::type> side;
    side.bias = a.bias; side.sigma = a.sigma; side.ib0 = ib0; side.I = a.I; side.with_rng = a.sample;
    side.prev_row = nullptr;
    side.prev_coh = (a.prev && j < a.J && ib0 < a.I) ? a.prev + (size_t)j * a.ldo + ib0 : nullptr;
    side.wfA = wt.wfA; side.wfB = wt.wfB; side.nA = wt.nA; side.nB = wt.nB;
    side.verdict = wt.verdict; side.stop = wt.stop; side.lim = wt.lim; side.total = wt.total;
    side.gen = gen; side.status = status; side.s_abort = s_abort; side.stamp = stamps ? stamps + 1 : nullptr;
#pragma unroll
    for (int e = 0; e < E; ++e) side.pv[e] = 0.f;
    const PhiloxKey key = a.key;
    side.rng.init(key, ((unsigned long long)(a.row0 + j) * (unsigned long long)a.I + ib0) >> 2);
    f32x4 acc[G::MI][1];
#pragma unroll
    for (int t = 0; t < G::MI; ++t) acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.acc_init && j < a.J) {                   // start the chain from a stored partial sum (written before the launch)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < G::MI; ++t) {
                const int i = ib0 + G::MI * r + t;
                if (i < a.I) acc[t][0][r] = a.acc_init[(size_t)j * a.ld_init + i];
            }
    }
    mainloop<XM, G, true, SEG2, 0, PL, STG_DMA>(acc, kr, i0, j0, smem, side);
    aborted = side.aborted;
    // the next claim goes out under the epilogue (bm_chain.h: a claim binds a tile to this workgroup)
    // ... together with a look at the stop word: the main loop's cheap test before the next tile starts
    if (tid == 0) {
        nxt = __hip_atomic_fetch_add(claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stop_seen = __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ... and the arrival words the NEXT mean-field tile of this workgroup will ask for (its sweep is this position's or
        // the next): they cross the fabric while the epilogue runs instead of at the head of that tile's wait
        pre0 = __hip_atomic_load(pre_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pre1 = __hip_atomic_load(pre_ptr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    dmax = 0.f;
    if (stamps && tid == 0) stamps[2] = wall_clock64();
    if (!aborted) dmax = act_epilogue<G, 0>(a, key, acc, side, i0, j0);
    if (stamps && tid == 0) stamps[3] = wall_clock64();
}

// c.dbg: measurements only - 2: no waits (WRONG results)
__global__ __launch_bounds__(GeoChain::NT, 1) void dbm_chain_kernel(DchArgs c) {
    using G = GeoChain;
    __shared__ __attribute__((aligned(16))) float smem[G::SMEM_FLOATS];
    __shared__ unsigned s_ctl[16];         // [0] next claim of the main sequence, [1] particle claim, [2] abort, [3] stop word, [4..11] wave residuals, [12..13] prefetched arrival words
    const int tid = threadIdx.x;
    unsigned xcc_reg;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_reg));
    const int team = (int)(xcc_reg & 7u);
    if ((xcc_reg & 0xfu) > 7u) {
        if (tid == 0) atomicExch(c.status, CHAIN_ERR_XCC);
        return;
    }
    if (blockIdx.x == 0 && tid < 16) c.claim_zero[(tid >> 1) * 32 + (tid & 1) * 16] = 0u;
    unsigned *cl_main = c.claim + team * 32, *cl_pc = cl_main + 16;
    const bool mf_team = c.mf_sweeps > 0 && team < c.tiles_mf && !*c.done0;
    const bool pc_team = c.pc_sweeps > 0 && team < c.tiles_pc;
    const int n1 = c.mf[0].ntile, n2 = c.mf[1].ntile, S = c.slots, T = n1 + S + n2;
    const int q0 = c.pc[0].ntile, q1 = q0 + c.pc[1].ntile, q2 = q1 + c.pc[2].ntile;     // particle tiles per sweep: [h1 | h2 | v]
    const int npc = pc_team ? c.pc_sweeps * q2 : 0;
    bool tail = !mf_team;
    if (tid == 0) {
        s_ctl[0] = tail ? 0u : __hip_atomic_fetch_add(cl_main, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ctl[1] = tail ? __hip_atomic_fetch_add(cl_pc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
    __syncthreads();
    // (everything read back from LDS is workgroup-uniform; saying so keeps the pass templates - indexed by it - in scalar
    //  registers: the DMA instructions take their chunk bases from SGPRs)
#define DCH_UNI(x) __builtin_amdgcn_readfirstlane((int)(x))
    int n = DCH_UNI(s_ctl[0]);      // pending claim of the main sequence (main mode)
    int p = DCH_UNI(s_ctl[1]);      // pending particle claim (tail mode, or behind a slot of the main sequence)
    unsigned stop_word = 0u;        // the stop word as last seen under a tile's epilogue
    unsigned seen0 = 0u, seen1 = 0u;                // arrived[seen_base], arrived[seen_base + 1] as seen under the last epilogue
    int seen_base = -1;
    int nstamp = 0;
    int pos_sweep = 1;              // sweep of this workgroup's position in the main sequence
    while (true) {
        int fam, sweep = 0, pass = 0, ti = 0;
        bool from_main = false;
        if (!tail) {
            const int s = small_div(n, T), r = n - s * T;
            sweep = s + 1;
            pos_sweep = sweep <= c.mf_sweeps ? sweep : c.mf_sweeps + 1;
            bool ended = sweep > c.mf_sweeps;
            if (!ended && sweep >= 3)          // what the last epilogue saw; the wait of the tile itself is the safety net
                ended = dch_stopped(stop_word, (unsigned)(sweep - 2));
            if (ended) {
                tail = true;
                if (tid == 0) s_ctl[1] = __hip_atomic_fetch_add(cl_pc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                p = DCH_UNI(s_ctl[1]);
                __syncthreads();
                continue;
            }
            from_main = true;
            if (r >= n1 && r < n1 + S) {       // a slot for a particle tile
                if (tid == 0) s_ctl[1] = npc > 0 ? __hip_atomic_fetch_add(cl_pc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffffu;
                __syncthreads();
                p = DCH_UNI(s_ctl[1]);
                __syncthreads();
                if (p >= npc) {                // none left: next entry of the main sequence
                    if (tid == 0) s_ctl[0] = __hip_atomic_fetch_add(cl_main, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __syncthreads();
                    n = DCH_UNI(s_ctl[0]);
                    __syncthreads();
                    continue;
                }
                fam = 1;
            } else {
                fam = 0;
                pass = r < n1 ? 0 : 1;
                ti = r < n1 ? r : r - n1 - S;
            }
        } else {
            if (p >= npc) break;
            fam = 1;
        }
        ActArgs a;
        a.P2 = Operand{nullptr, 0, 0, 0}; a.Q2 = a.P2; a.K2 = 0;
        a.negmeans = nullptr; a.maxdiff = nullptr; a.maxdiff_blk = nullptr;
        a.rowacc = nullptr; a.beta_a = 0.f; a.beta_b = 0.f; a.rowacc_single = 0; a.rowdot_out = nullptr; a.ld_part = 0;
        a.dot_vec = nullptr; a.dot_mat = nullptr; a.ld_dot = 0;
        a.acc_init = nullptr; a.ld_init = 0; a.skip = nullptr; a.prev = nullptr;
        a.chk_ctl = nullptr; a.chk_slots = nullptr; a.chk_n = 0; a.chk_tol = 0.f;
        a.b3 = Bf3Range{}; a.states16 = nullptr; a.ld16 = 0; a.map_xi = 0;
        a.fe_rowacc2 = nullptr; a.fe_flip = nullptr; a.fe_x = nullptr; a.fe_ldx = 0; a.fe_w = nullptr; a.fe_ldw = 0;
        a.fe_zero = nullptr; a.fe_rm = 0; a.fe_key = PhiloxKey{0u, 0u, 0u, 0u};
        a.mult = 1.f; a.bmult = 1.f;
#ifdef BM_PROBE
        a.dbg = nullptr;
#endif
        DchWait wt;
        wt.wfA = nullptr; wt.wfB = nullptr; wt.nA = 0; wt.nB = 0; wt.verdict = nullptr; wt.stop = nullptr; wt.lim = 0u;
        wt.total = (unsigned)c.arrive_total;
        unsigned *flag_out;
        bool seg2 = false;
        if (fam == 0) {
            const DchPass &ph = c.mf[pass];
            const int bw = sweep % 3, br = (sweep + 2) % 3;              // buffers written / read by this sweep
            a.P1 = ph.P1; a.K1 = ph.K1; a.p_xm = ph.p_xm; a.I = ph.I; a.J = c.J_mf;
            a.bias = ph.bias; a.sigma = ph.sigma; a.kind = ph.kind; a.sample = 0;
            a.key = ph.key; a.row0 = 0;
            unsigned *fl = c.flags_mf + (size_t)team * (2 * DCH_MAXSW) * DCH_MAXTI;
            const int pm = 2 * (sweep - 1) + pass;
            if (pass == 0) {
                a.Q1 = make_operand_dev(c.mu[br][1], c.ld_mu[1], c.J_mf);
                a.acc_init = c.xw0; a.ld_init = c.ld_xw0;
                a.means = c.mu[bw][0]; a.ldo = c.ld_mu[0]; a.prev = c.mu[br][0];
                if (sweep > 1) { wt.wfA = fl + (size_t)(pm - 1) * DCH_MAXTI; wt.nA = n2; }
            } else {
                a.Q1 = make_operand_dev(c.mu[bw][0], c.ld_mu[0], c.J_mf);
                a.means = c.mu[bw][1]; a.ldo = c.ld_mu[1]; a.prev = c.mu[br][1];
                wt.wfA = fl + (size_t)(pm - 1) * DCH_MAXTI; wt.nA = n1;
            }
            a.states = nullptr;
            if (sweep >= 3) {
                wt.verdict = c.arrived + (sweep - 2); wt.stop = c.stop; wt.lim = (unsigned)(sweep - 2);
                // already seen complete under the last epilogue?  (a complete word is final; "converged" is handled by the
                // wait: it aborts the tile and raises the stop word)
                const unsigned sv = (sweep - 2 == seen_base) ? seen0 : ((sweep - 2 == seen_base + 1) ? seen1 : 0u);
                if ((sv & 0xffffu) == (unsigned)c.arrive_total && (sv >> 16) != 0u) wt.verdict = nullptr;
            }
            flag_out = fl + (size_t)pm * DCH_MAXTI + ti;
        } else {
            const int t = small_div(p, q2), r = p - t * q2;
            pass = r < q0 ? 0 : (r < q1 ? 1 : 2);
            ti = r < q0 ? r : (r < q1 ? r - q0 : r - q1);
            const DchPass &ph = c.pc[pass];
            const int bi = t & 1, bo = bi ^ 1;                            // particle buffers read / written by sweep t
            a.P1 = ph.P1; a.K1 = ph.K1; a.p_xm = ph.p_xm; a.I = ph.I; a.J = c.J_pc;
            a.bias = ph.bias; a.sigma = ph.sigma; a.kind = ph.kind; a.sample = ph.sample;
            a.key = ph.key; a.key.site += 16u * (unsigned)t; a.row0 = c.prow0;
            unsigned *fl = c.flags_pc + (size_t)team * (3 * DCH_MAXPC) * DCH_MAXTI;
            const int pq = 3 * t + pass;
            float *out; int ldo;
            if (pass == 0) {        // h1 <- v (segment 1) + h2 (segment 2), both of the sweep before
                a.Q1 = make_operand_dev(c.pv[bi], c.ld_v, c.J_pc);
                a.P2 = ph.P2; a.K2 = ph.K2; a.Q2 = make_operand_dev(c.ph[bi][1], c.ld_h[1], c.J_pc);
                seg2 = true;
                out = c.ph[bo][0]; ldo = c.ld_h[0];
                if (t > 0) {
                    wt.wfA = fl + (size_t)(pq - 1) * DCH_MAXTI; wt.nA = c.pc[2].ntile;       // v of sweep t-1
                    wt.wfB = fl + (size_t)(pq - 2) * DCH_MAXTI; wt.nB = c.pc[1].ntile;       // h2 of sweep t-1
                }
            } else if (pass == 1) { // h2 <- new h1
                a.Q1 = make_operand_dev(c.ph[bo][0], c.ld_h[0], c.J_pc);
                out = c.ph[bo][1]; ldo = c.ld_h[1];
                wt.wfA = fl + (size_t)(pq - 1) * DCH_MAXTI; wt.nA = c.pc[0].ntile;
            } else {                // v <- new h1
                a.Q1 = make_operand_dev(c.ph[bo][0], c.ld_h[0], c.J_pc);
                out = c.pv[bo]; ldo = c.ld_v;
                wt.wfA = fl + (size_t)(pq - 2) * DCH_MAXTI; wt.nA = c.pc[0].ntile;
            }
            a.ldo = ldo;
            a.means = ph.sample ? nullptr : out;       // without sampling the layer's value is its mean
            a.states = ph.sample ? out : nullptr;
            flag_out = fl + (size_t)pq * DCH_MAXTI + ti;
        }
        if (c.dbg & 2) { wt.wfA = nullptr; wt.wfB = nullptr; }
        const int tj = team;                           // one row block per team and family (host: tiles <= 8)
        unsigned nxt = 0, stop_seen = 0, pre0 = 0, pre1 = 0;
        const int pre_base = pos_sweep >= 2 ? pos_sweep - 2 : 0;          // arrived[] has two words of slack at the end
        float dmax = 0.f;
        int aborted = 0;
        long long *stp = (c.stamps && nstamp < DCH_STAMP_TILES) ? c.stamps + ((size_t)blockIdx.x * DCH_STAMP_TILES + nstamp) * 8 : nullptr;
        if (stp && tid == 0) { stp[0] = wall_clock64(); stp[5] = (long long)(fam * 1000 + pass * 100 + (fam ? p / q2 : sweep)); stp[6] = (long long)ti; stp[7] = (long long)team; }
        ++nstamp;
        // the claim that goes out under this tile's epilogue: the main sequence in main mode, the particle counter in tail mode
        unsigned *cl_next = tail ? cl_pc : cl_main;
        if (seg2)          dch_tile<G, true, KM>(a, ti * G::TI, tj * G::TJ, smem, wt, c.gen, c.status, s_ctl + 2, cl_next, nxt, c.stop, stop_seen, c.arrived + pre_base, pre0, pre1, dmax, aborted, stp);
        else if (!a.p_xm)  dch_tile<G, false, KM>(a, ti * G::TI, tj * G::TJ, smem, wt, c.gen, c.status, s_ctl + 2, cl_next, nxt, c.stop, stop_seen, c.arrived + pre_base, pre0, pre1, dmax, aborted, stp);
        else               dch_tile<G, false, XM>(a, ti * G::TI, tj * G::TJ, smem, wt, c.gen, c.status, s_ctl + 2, cl_next, nxt, c.stop, stop_seen, c.arrived + pre_base, pre0, pre1, dmax, aborted, stp);
        // publish: every wave's stores are in the L2, then the tile's flag; the barrier also frees the LDS ring
        if (fam == 0) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));
            if ((tid & 63) == 0) s_ctl[4 + (tid >> 6)] = __float_as_uint(dmax);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) { s_ctl[tail ? 1 : 0] = nxt; s_ctl[3] = stop_seen; s_ctl[12] = pre0; s_ctl[13] = pre1; }
        wg_barrier();
        if (tid == 0 && !aborted) {
            __hip_atomic_store(flag_out, c.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fam == 0) {
                unsigned m = 0u;                       // residuals are >= 0: their bit patterns order like the values
#pragma unroll
                for (int q = 0; q < G::NT / 64; ++q) m = s_ctl[4 + q] > m ? s_ctl[4 + q] : m;
                // the loop goes on while SOME element moved by more than tol (dbm.py:449-452): count the tiles that saw one.
                // Fire and forget: waiting for a returned atomic here held all 8 waves for ~3.5 us behind every tile.
                (void)__hip_atomic_fetch_add(c.arrived + sweep, 1u + ((__uint_as_float(m) > c.tol) ? 0x10000u : 0u),
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (stp && tid == 0) stp[4] = wall_clock64();
        if (tail) p = DCH_UNI(s_ctl[1]); else n = DCH_UNI(s_ctl[0]);
        stop_word = (unsigned)DCH_UNI(s_ctl[3]);
        seen0 = (unsigned)DCH_UNI(s_ctl[12]); seen1 = (unsigned)DCH_UNI(s_ctl[13]); seen_base = pre_base;
        (void)from_main;
        wg_barrier();                                  // s_ctl is rewritten by the next round
    }
#undef DCH_UNI
}

// after the launch: the trip count (dbm.py:449-457) from the per-sweep residuals; steps = the first sweep whose residual
// is <= tol, or max_mf_updates
__global__ void dch_finish_kernel(MfCtl *ctl, const unsigned *arrived, int total, int mf_sweeps) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (ctl->done) return;                             // the step-0 condition ended the loop: steps stays 0
    int n = mf_sweeps, conv = 0;
    for (int s = 1; s <= mf_sweeps; ++s) {
        if ((int)(arrived[s] & 0xffffu) != total) { n = -s; break; }   // cannot happen before convergence: reported by the host
        if ((arrived[s] >> 16) == 0u) { n = s; conv = 1; break; }
    }
    ctl->steps = n;
    ctl->done = conv || n == mf_sweeps;
}

}  // namespace bm
