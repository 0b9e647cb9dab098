// bm_chain.h — a run of DEPENDENT propagation passes (h0 <- v0, v1 <- h0, h1 <- v1, ...: base_rbm.py:367-378, :417-426)
// as ONE launch instead of one launch per pass.
//
// Why: at the north-star shape a pass is 5 us of matrix work inside a 12 - 14 us kernel, and ~4 us of that is the
// kernel boundary (2.3 us with no wave running, 0.4 us of prologue, 1.3 us until the first chunks arrive from an L2 that
// every boundary invalidates).  What a pass needs from its predecessor is row-local, though: output rows
// [j0, j0 + 64) of pass p+1 read rows [j0, j0 + 64) of pass p's output and nothing else.  So the rows are dealt out to the
// 8 XCDs - batch row block tj belongs to TEAM tj % 8 = the workgroups running on XCD tj % 8 - and the hand-over from pass
// to pass stays inside ONE L2:
//   * a workgroup learns its XCD from the hardware (HW_REG_XCC_ID), not from its block index: "my team mates share my
//     L2" is then true by construction;
//   * work is CLAIMED, not assigned: a team's tiles, ordered (row block round, pass, tile column), are handed out by one
//     atomic counter per team, so any number >= 1 of resident workgroups per XCD makes progress, and a workgroup that
//     claimed tile n only ever waits for tiles < n, all of which are claimed by running workgroups: no deadlock;
//   * producer: stores, `s_waitcnt vmcnt(0)` (the L2 has acknowledged them), workgroup barrier, then ONE flag word per
//     tile (store sc1) holding the launch generation - flags are never reset;
//   * consumer: requests its weight panel chunks (constant during the launch) FIRST, then one wave polls the <= 64 flags
//     of the producing pass with one sc1 load per lane, then the activation chunks are requested with sc1 loads (past the
//     CU's vector L1, which may still hold last sweep's contents of those lines; served by the XCD's L2).
//     tools/teamprobe.hip measured this hand-over at ~1.6 us against ~3.6 us for a kernel boundary, with every value
//     checked over ping-pong buffers; agent-scope fences (L2 writeback + invalidate per workgroup) cost 30 us per pass
//     (tools/chainprobe.hip) and are not used: nothing here needs another XCD to see the data before the kernel ends.
//   * the weights stay in the XCD's L2 from pass to pass (each XCD reads all of W: 3.2 MB of 4 MB at 784 x 1024).
// Results are bit-identical to the per-pass launches: same tile body (mainloop / act_epilogue), same canonical order,
// same Philox addressing (global row, column); only WHERE and WHEN a tile runs changes.
#pragma once
#include "bm_kernels.h"

namespace bm {

constexpr int CHAIN_MAXPH = 24;       // passes per launch (the 4 KiB kernel-argument budget: 152 bytes per pass); longer runs take several launches
constexpr int CHAIN_MAXTI = 64;       // tile columns per pass (one poll = one load per lane)
using GeoChain = GeoAct8;             // 32 x 64 tile, 8 waves: the tuner's choice for the shapes this path serves

// the fields of ActArgs a plain RBM pass uses
struct ChainPhase {
    Operand P, Q; int K; int p_xm; int I;
    const float *bias, *sigma;
    float mult, bmult; int kind, sample;
    float *means, *states, *negmeans; int ldo;
    PhiloxKey key; long long row0;
};

struct ChainArgs {
    int nph, J, tiles_j, rounds_cap, dbg;
    unsigned gen;
    unsigned *flags;          // [8 teams][rounds_cap][CHAIN_MAXPH][CHAIN_MAXTI]: generation of the launch that completed the tile
    unsigned *claim;          // [8][32] (128-byte spacing): this launch's claim counters, zero at launch
    unsigned *claim_zero;     // the counters a LATER launch will use: zeroed here by workgroup 0
    int *status;              // [0] != 0: a wait expired (sticky; results invalid)   [1] tiles computed (all launches)
    long long *stamps;        // developer timeline (BM355_DEBUG=chain_stamps=file): [block][16 tiles][8] 100 MHz clock values, else null
    int tp0[CHAIN_MAXPH + 1]; // prefix sums of the tile columns per pass
    ChainPhase ph[CHAIN_MAXPH];
};

static_assert(sizeof(ChainArgs) <= 4096, "kernel arguments");
// (tried: the passes' arguments staged in LDS instead of fetched from the kernel-argument segment with a run-time offset,
//  and the Philox blocks computed after the Q requests instead of before the wait: 26.0 against 25.3 us per sweep)
enum : int { CHAIN_ERR_TIMEOUT = 1, CHAIN_ERR_XCC = 2 };

template <int E, class Rng, bool COH> struct ChainSide : ActSide<E, Rng> {
    static constexpr bool kSplitFill = true, kCohQ = COH, kCanAbort = false;
    const unsigned *wflags;      // flags of the producing pass for this row block (null: pass 0 reads launch inputs)
    int nwait;
    unsigned gen;
    int *status;
    long long *stamp;            // null, or where the end of the wait is recorded
    int nosleep;
    // one wave polls (lane = producing tile), everybody meets at the barrier.  Bounded: an expired wait sets the sticky
    // status word and lets every later wait fall through (wrong results, reported by bm_rbm_sync; no hang).
    __device__ __forceinline__ void wait_inputs() {
        if (!wflags) { if (stamp && threadIdx.x == 0) *stamp = wall_clock64(); return; }      // wave-uniform
        if ((threadIdx.x >> 6) == 0) {
            const int lane = threadIdx.x & 63;
            unsigned spins = 0;
            for (;;) {
                unsigned v = gen;
                if (lane < nwait) v = __hip_atomic_load(wflags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(v == gen)) break;
                if (!nosleep) __builtin_amdgcn_s_sleep(1);
                if ((++spins & 1023u) == 0) {
                    const int st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (st || spins > (1u << 21)) {
                        if (lane == 0 && !st) atomicExch(status, CHAIN_ERR_TIMEOUT);
                        break;
                    }
                }
            }
        }
        wg_barrier();
        if (stamp && threadIdx.x == 0) *stamp = wall_clock64();
    }
};

// one output tile of one pass (the middle of act_kernel)
template <class G, int PL, bool COH>
__device__ __forceinline__ void chain_tile(const ActArgs &a, int i0, int j0, float *smem, const unsigned *wflags, int nwait,
                                           unsigned gen, int *status, long long *stamps, int nosleep, unsigned *claim, unsigned &nxt) {
    constexpr int E = G::E;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w % G::WI, wj = w / G::WI;
    const int g = lane >> 4, l15 = lane & 15;
    const int ib0 = i0 + wi * (16 * G::MI) + g * E;
    const int j = j0 + wj * 16 + l15;
    KRange kr;
    kr.P1 = a.P1; kr.Q1 = a.Q1; kr.K1 = a.K1;
    kr.P2 = a.P2; kr.Q2 = a.Q2; kr.K2 = 0;
    ChainSide<E, typename PhiloxFor<G::MI>::type, COH> side;
    side.bias = a.bias; side.sigma = a.sigma; side.ib0 = ib0; side.I = a.I; side.with_rng = a.sample;
    side.prev_row = nullptr;
    side.wflags = wflags; side.nwait = nwait; side.gen = gen; side.status = status; side.stamp = stamps ? stamps + 1 : nullptr; side.nosleep = nosleep;
    const PhiloxKey key = a.key;
    side.rng.init(key, ((unsigned long long)(a.row0 + j) * (unsigned long long)a.I + ib0) >> 2);
    f32x4 acc[G::MI][1];
#pragma unroll
    for (int t = 0; t < G::MI; ++t) acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mainloop<XM, G, true, false, 0, PL, STG_DMA>(acc, kr, i0, j0, smem, side);
    // The next claim goes out HERE, under the epilogue - not at the start of the tile: a claim binds a tile to this
    // workgroup, and a tile of the CURRENT pass bound to a workgroup that is still a whole tile away from starting it
    // leaves idle team mates (who hold later passes' tiles) waiting for it - measured: every pass took two rounds.
    if (tid == 0) nxt = __hip_atomic_fetch_add(claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (stamps && tid == 0) stamps[2] = wall_clock64();
    (void)act_epilogue<G, 0>(a, key, acc, side, i0, j0);
    if (stamps && tid == 0) stamps[3] = wall_clock64();
}

// c.dbg (BM355_DEBUG=chain_dbg): measurements only - 2: no waits (WRONG results: prices the hand-overs), 4: poll without s_sleep.
// COH = false (plain Q loads) measured the same time as sc1 loads and is not instantiated.
template <bool COH>
__global__ __launch_bounds__(GeoChain::NT, 1) void act_chain_kernel(ChainArgs c) {
    using G = GeoChain;
    __shared__ __attribute__((aligned(16))) float smem[G::SMEM_FLOATS];
    __shared__ unsigned s_claim;
    const int tid = threadIdx.x;
    unsigned xcc_reg;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_reg));
    const int team = (int)(xcc_reg & 7u);
    if ((xcc_reg & 0xfu) > 7u) {                       // not an 8-XCD part: the host gate should have kept us away
        if (tid == 0) atomicExch(c.status, CHAIN_ERR_XCC);
        return;
    }
    if (blockIdx.x == 0 && tid < 8) c.claim_zero[tid * 32] = 0u;
    unsigned *my_claim = c.claim + team * 32;
    if (tid == 0) s_claim = __hip_atomic_fetch_add(my_claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    int n = (int)s_claim;
    const int T = c.tp0[c.nph];
    const int R = (c.tiles_j - team + 7) >> 3;          // row blocks of this team: team, team + 8, ...
    int ntiles = 0;
    while (true) {
        const int r = small_div(n, T);
        if (r >= R) break;                              // uniform
        const int rem = n - r * T;
        int p = 0;
#pragma unroll
        for (int q = 1; q < CHAIN_MAXPH; ++q) p += (q < c.nph && rem >= c.tp0[q]) ? 1 : 0;
        const int ti = rem - c.tp0[p];
        unsigned nxt = 0;
        const ChainPhase &ph = c.ph[p];
        ActArgs a;
        a.P1 = ph.P; a.Q1 = ph.Q; a.K1 = ph.K; a.p_xm = ph.p_xm;
        a.P2 = Operand{nullptr, 0, 0, 0}; a.Q2 = a.P2; a.K2 = 0;
        a.I = ph.I; a.J = c.J;
        a.bias = ph.bias; a.sigma = ph.sigma; a.mult = ph.mult; a.bmult = ph.bmult; a.kind = ph.kind; a.sample = ph.sample;
        a.means = ph.means; a.states = ph.states; a.negmeans = ph.negmeans; a.ldo = ph.ldo;
        a.key = ph.key; a.row0 = ph.row0;
        a.prev = nullptr; a.maxdiff = nullptr; a.maxdiff_blk = nullptr;
        a.rowacc = nullptr; a.beta_a = 0.f; a.beta_b = 0.f; a.rowacc_single = 0; a.rowdot_out = nullptr; a.ld_part = 0;
        a.dot_vec = nullptr; a.dot_mat = nullptr; a.ld_dot = 0;
        a.acc_init = nullptr; a.ld_init = 0; a.skip = nullptr;
        a.chk_ctl = nullptr; a.chk_slots = nullptr; a.chk_n = 0; a.chk_tol = 0.f;
        a.b3 = Bf3Range{}; a.states16 = nullptr; a.ld16 = 0; a.map_xi = 0;
        a.fe_rowacc2 = nullptr; a.fe_flip = nullptr; a.fe_x = nullptr; a.fe_ldx = 0; a.fe_w = nullptr; a.fe_ldw = 0;
        a.fe_zero = nullptr; a.fe_rm = 0; a.fe_key = PhiloxKey{0u, 0u, 0u, 0u};
#ifdef BM_PROBE
        a.dbg = nullptr;
#endif
        unsigned *fl = c.flags + ((size_t)(team * c.rounds_cap + r) * CHAIN_MAXPH) * CHAIN_MAXTI;
        const unsigned *wfl = (p > 0 && !(c.dbg & 2)) ? fl + (size_t)(p - 1) * CHAIN_MAXTI : nullptr;
        const int nwait = p > 0 ? c.tp0[p] - c.tp0[p - 1] : 0;
        const int tj = team + 8 * r;
        long long *stp = c.stamps ? c.stamps + ((size_t)blockIdx.x * 16 + (ntiles < 15 ? ntiles : 15)) * 8 : nullptr;
        if (stp && tid == 0) { stp[0] = wall_clock64(); stp[5] = (long long)p; stp[6] = (long long)ti; stp[7] = (long long)team; }
        if (ph.p_xm) chain_tile<G, XM, COH>(a, ti * G::TI, tj * G::TJ, smem, wfl, nwait, c.gen, c.status, stp, c.dbg & 4, my_claim, nxt);
        else         chain_tile<G, KM, COH>(a, ti * G::TI, tj * G::TJ, smem, wfl, nwait, c.gen, c.status, stp, c.dbg & 4, my_claim, nxt);
        // publish: every wave's stores are in the L2, then the tile's flag; the barrier also frees the LDS ring
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) s_claim = nxt;
        wg_barrier();
        if (tid == 0) __hip_atomic_store(fl + (size_t)p * CHAIN_MAXTI + ti, c.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (stp && tid == 0) stp[4] = wall_clock64();
        n = (int)s_claim;
        ++ntiles;
    }
    if (tid == 0 && ntiles) atomicAdd(c.status + 1, ntiles);
}

// ---- host side: the recorder the RBM entry points wrap around a run of passes
struct ChainState {
    unsigned *flags = nullptr, *claim = nullptr;     // claim: ring of CLAIM_SLOTS x [8][32]
    int *status = nullptr;                           // device [2]
    int rounds_cap = 0;
    unsigned gen = 0;
    unsigned launches = 0;
    long long tiles_expected = 0;                    // what status[1] must read when the stream is idle
    bool on = false;                                 // recording
    int mode = -1;                                   // BM355_DEBUG=chain: 0 off, 1 auto (default), 2 force where legal
    int ncu = 0;
    std::vector<ActArgs> rec;
    // per recorded pass: an x-major P operand (W^T) the PER-PASS launch should use instead of rec[i].P1 (ptr null: none).
    // A chained launch keeps the k-major W: with both W and W^T streaming through an XCD's 4 MiB L2 neither stays
    // resident (measured: 26.0 against 25.2 us per sweep).
    std::vector<Operand> alt_p;
    long long *stamps = nullptr;                     // BM355_DEBUG=chain_stamps=file: timeline of the LAST chained launch, dumped at release
    static constexpr int CLAIM_SLOTS = 16;
    static constexpr size_t STAMP_WORDS = 256 * 16 * 8;
    void release() {
        if (stamps) {
            const char *f = bm::dbg("chain_stamps");
            std::vector<long long> hst(STAMP_WORDS);
            if (f && hipMemcpy(hst.data(), stamps, STAMP_WORDS * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                FILE *fp = fopen(f, "wb");
                if (fp) { fwrite(hst.data(), 8, STAMP_WORDS, fp); fclose(fp); }
            }
            (void)hipFree(stamps); stamps = nullptr;
        }
        if (flags) (void)hipFree(flags);
        if (claim) (void)hipFree(claim);
        if (status) (void)hipFree(status);
        flags = claim = nullptr; status = nullptr;
    }
};

static inline int chain_mode(ChainState &cs) {
    if (cs.mode < 0) {
        const char *e = bm::dbg("chain");
        cs.mode = e ? atoi(e) : 1;
        hipDeviceProp_t pr; int d = 0, nxcc = 0; (void)hipGetDevice(&d);
        const bool have = hipGetDeviceProperties(&pr, d) == hipSuccess;
        cs.ncu = have ? pr.multiProcessorCount : 0;
        if (hipDeviceGetAttribute(&nxcc, hipDeviceAttributeNumberOfXccs, d) != hipSuccess) nxcc = 0;
        // teams are XCDs: a gfx950 / gfx942 part with 8 XCDs in single-partition mode (MI355X SPX: 256 CUs, round-robin
        // dispatch over the XCDs, one 4 MiB L2 each).  Anything else keeps its per-pass launches.
        const bool arch = have && (strncmp(pr.gcnArchName, "gfx950", 6) == 0 || strncmp(pr.gcnArchName, "gfx942", 6) == 0);
        if (!(arch && nxcc == 8 && cs.ncu == 256)) cs.mode = 0;
    }
    return cs.mode;
}

// may this pass run as a phase of a chained launch?
static inline bool chain_phase_ok(const ActArgs &a) {
    using G = GeoChain;
    if (a.K2 > 0 || a.b3.K1 > 0 || a.prev || a.maxdiff || a.maxdiff_blk || a.rowacc || a.rowdot_out || a.acc_init ||
        a.skip || a.chk_ctl || a.states16 || a.dot_mat || a.fe_flip) return false;
    if (a.kind != 0 && a.kind != 1) return false;
    const int pl = a.p_xm ? XM : KM;
    if (!(operand_fast(a.P1, pl, a.K1) && operand_fast(a.Q1, XM, a.K1))) return false;
    if (a.K1 < G::PF * G::BK) return false;                       // split fill: chunks 0 .. PF-1 are full DMA chunks
    if ((a.I + G::TI - 1) / G::TI > CHAIN_MAXTI) return false;
    return true;
}

// Launch the recorded passes: chained where the run qualifies, one launch per pass otherwise.
static inline int chain_flush(ChainState &cs, hipStream_t st, int maxB) {
    cs.on = false;
    std::vector<ActArgs> rec;
    std::vector<Operand> alt;
    rec.swap(cs.rec);
    alt.swap(cs.alt_p);
    if (rec.empty()) return 0;
    alt.resize(rec.size(), Operand{nullptr, 0, 0, 0});
    using G = GeoChain;
    // default rule: six passes or more (CD-k with k >= 3, three sampling sweeps).  A chained pass costs ~12.3 us against
    // 12.9 / 14.1 us for the per-pass kernels, but the launch pays ~3 us up front and its first pass runs with every XCD
    // pulling ALL of W through the fabric instead of an eighth: the three passes of a CD-1 update take 42.9 us chained
    // against 39.9 us as three launches (same box, rocprofv3 kernel trace; 67.4 against 64.0 us per update), twenty
    // passes 253 against 283 us.
    bool ok = chain_mode(cs) > 0 && rec.size() >= (cs.mode == 1 ? 6u : 2u);
    const int J = rec[0].J, tiles_j = (J + G::TJ - 1) / G::TJ;
    for (const ActArgs &a : rec) ok = ok && chain_phase_ok(a) && a.J == J;
    // auto mode: every XCD must own a row block, and a pass should be about one tile per CU (larger outputs keep their
    // CUs busy across a kernel boundary anyway, and their weights do not fit an L2)
    if (ok && cs.mode == 1) {
        ok = tiles_j >= 8;
        for (const ActArgs &a : rec) ok = ok && ((a.I + G::TI - 1) / G::TI) * ((tiles_j + 7) / 8) <= 2 * 32;
    }
    if (!ok) {
        for (size_t i = 0; i < rec.size(); ++i) {
            if (alt[i].ptr) { rec[i].P1 = alt[i]; rec[i].p_xm = 1; }
            launch_act(rec[i], st);
        }
        return 0;
    }
    const int rounds = (tiles_j + 7) / 8;
    if (!cs.flags || rounds > cs.rounds_cap) {
        if (cs.flags) { if (hipStreamSynchronize(st) != hipSuccess) return -1; (void)hipFree(cs.flags); cs.flags = nullptr; }
        const int cap = std::max(rounds, ((maxB + G::TJ - 1) / G::TJ + 7) / 8);
        const size_t nfl = (size_t)8 * cap * CHAIN_MAXPH * CHAIN_MAXTI;
        if (hipMalloc((void **)&cs.flags, nfl * 4) != hipSuccess) return -1;
        if (hipMemsetAsync(cs.flags, 0, nfl * 4, st) != hipSuccess) return -1;
        cs.rounds_cap = cap;
        cs.gen = 0;
        if (!cs.claim) {
            if (hipMalloc((void **)&cs.claim, (size_t)ChainState::CLAIM_SLOTS * 8 * 32 * 4) != hipSuccess) return -1;
            if (hipMemsetAsync(cs.claim, 0, (size_t)ChainState::CLAIM_SLOTS * 8 * 32 * 4, st) != hipSuccess) return -1;
            if (hipMalloc((void **)&cs.status, 2 * sizeof(int)) != hipSuccess) return -1;
            if (hipMemsetAsync(cs.status, 0, 2 * sizeof(int), st) != hipSuccess) return -1;
        }
    }
    for (size_t first = 0; first < rec.size(); first += CHAIN_MAXPH) {
        const int nph = (int)std::min<size_t>(CHAIN_MAXPH, rec.size() - first);
        ChainArgs c;
        memset(&c, 0, sizeof(c));
        c.nph = nph; c.J = J; c.tiles_j = tiles_j; c.rounds_cap = cs.rounds_cap;
        if (++cs.gen == 0) {                              // generation wrap: start over with clean flags
            if (hipMemsetAsync(cs.flags, 0, (size_t)8 * cs.rounds_cap * CHAIN_MAXPH * CHAIN_MAXTI * 4, st) != hipSuccess) return -1;
            cs.gen = 1;
        }
        c.gen = cs.gen;
        c.flags = cs.flags;
        c.claim = cs.claim + (size_t)(cs.launches % ChainState::CLAIM_SLOTS) * 8 * 32;
        c.claim_zero = cs.claim + (size_t)((cs.launches + ChainState::CLAIM_SLOTS / 2) % ChainState::CLAIM_SLOTS) * 8 * 32;
        ++cs.launches;
        c.status = cs.status;
        int tiles = 0;
        for (int p = 0; p < nph; ++p) {
            const ActArgs &a = rec[first + p];
            ChainPhase &ph = c.ph[p];
            ph.P = a.P1; ph.Q = a.Q1; ph.K = a.K1; ph.p_xm = a.p_xm; ph.I = a.I;
            ph.bias = a.bias; ph.sigma = a.sigma; ph.mult = a.mult; ph.bmult = a.bmult; ph.kind = a.kind; ph.sample = a.sample;
            ph.means = a.means; ph.states = a.states; ph.negmeans = a.negmeans; ph.ldo = a.ldo;
            ph.key = a.key; ph.row0 = a.row0;
            c.tp0[p] = tiles;
            tiles += (a.I + G::TI - 1) / G::TI;
        }
        for (int p = nph; p <= CHAIN_MAXPH; ++p) c.tp0[p] = tiles;
        cs.tiles_expected += (long long)tiles * tiles_j;
        // one workgroup per CU (96 KiB of LDS each), fewer when the whole launch has fewer tiles
        const int grid = std::min(cs.ncu, std::max(8, tiles * tiles_j));
        static const int dbg = bm::dbg("chain_dbg") ? atoi(bm::dbg("chain_dbg")) : 0;
        if (bm::dbg("chain_stamps") && !cs.stamps && hipMalloc((void **)&cs.stamps, ChainState::STAMP_WORDS * 8) != hipSuccess) cs.stamps = nullptr;
        if (cs.stamps) (void)hipMemsetAsync(cs.stamps, 0, ChainState::STAMP_WORDS * 8, st);
        c.stamps = cs.stamps;
        c.dbg = dbg;
        hipLaunchKernelGGL(act_chain_kernel<true>, dim3(grid), dim3(G::NT), 0, st, c);
    }
    return 0;
}

}  // namespace bm
