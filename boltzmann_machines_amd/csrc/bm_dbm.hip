// bm_dbm.hip — C-ABI entry points for the DBM path (include/bm355.h): block-Gibbs
// sweep, mean-field, PCD particles, train op, sample_v, reconstruction, AIS, ELBO.
//
// Reference graph restated: boltzmann_machines/dbm.py:385-427 (Gibbs sweep),
// :429-478 (mean-field), :480-509 (particles), :511-639 (train op + max-norm),
// :641-648 (sample_v), :650-736 (AIS), :738-759 (ELBO).  Every layer update is one
// act_kernel launch (two-sided inputs = two K segments of one MFMA pipeline).
#include "../../include/bm355.h"
#include "bm_common.h"
#include "bm_kernels.h"

#include <math.h>

using namespace bm;

struct bm_xchg;
// bm_xchg.hip: mf_resid_kernel + bm_xchg_allreduce_max1 + mf_latch_kernel as one launch (the per-sweep loop control of a
// data-parallel mean-field over the direct exchange)
static int xchg_mf_ctl_step(bm_xchg *x, MfCtl *ctl, float *blk, int nblk, float tol, int init, hipStream_t stream);

namespace {
// RNG sites (counter word 2 = site + 16 * sweep index); DESIGN.md "RNG"
enum : uint32_t { SITE_DBM_H = 8 /* + layer */, SITE_DBM_V = 12, SITE_AIS_X0 = 13 };
constexpr int MAXL = BM_DBM_MAX_LAYERS;
}  // namespace

struct bm_dbm {
    bm_dbm_config cfg;
    bm_xchg *xchg_used = nullptr;          // the direct exchange this engine last used (bm_dbm_sync checks its status word)
    // after bm_dbm_exchange_apply_direct (bm_xchg.hip) a rank holds only ITS column slices of the momentum buffers dW_i:
    // every reader - get_param, the single-GPU update, apply_step - fails until bm_dbm_exchange_gather_dw (check_dw)
    bool dw_sharded = false;
    unsigned dw_set_mask = 0;              // layers whose dW the host has replaced since the buffers became sharded
    int L, V, N, M;
    int n[MAXL + 1];                       // n[0] = V, n[i+1] = hidden layer i
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // The fantasy-particle sweeps (PCD) read only the parameters and the particles, the mean-field only the
    // parameters, X and mu: within one update they are independent, so the particle sweeps run on a second stream
    // (fork at the start of the update, join before the gradients) and fill the launch / fill / tail gaps of the
    // small mean-field kernels.  `cur` is the stream layer_update / gibbs_sweep enqueue on.
    hipStream_t stream2 = nullptr, cur = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int pcd_geo = 0;                               // tile of the particle passes while they share the chip with the mean-field
                                                   // loop (ActArgs::geo_hint; BM355_DEBUG=dbm_pcd_geo=N, 0 = the tuner's choice):
                                                   // 3 (32 x 32, 32 KiB) where there IS a loop of small latency-bound passes to
                                                   // share CUs with - two or more layers of <= 2M weights, <= 1024 rows: 1.388 ->
                                                   // 1.353 ms per update at 784-512-1024 x 512 - and the tuner's pick elsewhere
                                                   // (3072 x 5000, one layer: 1.21 -> 1.29 .. 1.51 ms with the small tile;
                                                   // profiles/r6_dbm_ab.txt)
    int pcd_geo_now = 0;                           // ... and only while the loop is not SHORT: the small tile makes the particle
                                                   // chain itself slower (228 against 190 us for PCD-5), and that chain is the
                                                   // critical path under a loop of 3 sweeps (0.384 against 0.364 ms per update
                                                   // with the hint; from ~6 sweeps on the hint wins: 0.409 against 0.437 ms at 7).
                                                   // Decided per update from the previous trip count: MORE than one loop sweep per
                                                   // particle sweep (BM355_DEBUG=dbm_pcd_ratio=N; the crossover measured at
                                                   // 784-512-1024 x 512, PCD-5 lies between 5 and 6 sweeps).
    int updates_seen = 0;                          // the first updates run on one stream (launch tuning measures alone)
    // mean-field loop control mirror: pinned host copies of `ctl`, one per enqueued group of sweeps, so that the next
    // group is enqueued BEFORE the previous group's result is read (the GPU never waits for the host)
    static constexpr int MF_RING = 4;
    MfCtl *ctl_host = nullptr;
    hipEvent_t ctl_ev[MF_RING] = {nullptr, nullptr, nullptr, nullptr};
    int mf_pred = 0;                               // trip count of the previous mean-field call (size of the first group)
    // variables
    Mat W[MAXL], Wt[MAXL], dW[MAXL];       // W[i]: [n[i]][n[i+1]]
    DevBuf vb, dvb, sigma;
    DevBuf hb[MAXL], dhb[MAXL], q[MAXL], mm[MAXL], pen[MAXL];
    Mat mu[MAXL], mu_alt[MAXL], mu_new[MAXL];      // [N][n[i+1]]: result, ping-pong partner, approx-inference init
    Mat v, v_new, H[MAXL], H_new[MAXL];            // particles [M][*]
    Mat recon;                                     // [N][V]
    DevBuf grad;                                   // data-parallel payload: [pos_i | neg_i per layer | sums]
    float *sums_p = nullptr;                       // column sums inside `grad`: [V | V | (n_i | n_i) per layer]
    size_t raw_off[MAXL][2];                       // offsets of the raw pos / neg outer products of layer i
    float (*mf_reduce)(float, void *) = nullptr;   // max over ranks of the mean-field residual through a HOST callback
    void *mf_ctx = nullptr;                        // (bm_dbm_set_mf_allreduce: collectives the library does not own)
    bm_comm *comm = nullptr;                       // bm_dbm_set_comm: the residual is all-reduced (max) ON DEVICE, on the
                                                   // engine stream, by the library's own RCCL communicator
    bm_xchg *xchg = nullptr;                       // bm_dbm_set_xchg: the same through the direct peer-memory exchange
    DevBuf wnorm[MAXL];
    DevBuf mn_fac[MAXL];                           // max-norm column factors [2][n_{i+1}]: min(norm, c) | max(norm, 1e-8)
    Mat logits[MAXL];                              // Multinomial layers: row store of the logits / means [rows][n_i], on demand
    int logit_rows[MAXL] = {0, 0, 0, 0};
    bool failed = false;                           // a launch helper could not allocate (sticky; reported by the entry points)
    bool multinomial(int layer) const { return layer >= 0 && cfg.h_unit[layer] == BM_UNIT_MULTINOMIAL; }
    unsigned *flag = nullptr;                      // mean-field residual cell (= &ctl->maxdiff)
    MfCtl *ctl = nullptr;                          // device-side loop control
    DevBuf mfblk;                                  // [2][MAXL * BM_MF_SLOTS] per-workgroup residual slots of the mean-field
                                                   // sweeps, double-buffered by sweep parity (ActArgs::chk_ctl)
    Mat xw0;                                       // [N][n1] hoisted X.W0 of the current minibatch
    double *scal = nullptr;
    // AIS / ELBO workspaces (allocated on demand)
    int ais_rows = 0;
    Mat ax, ax2, av, ah2;
    DevBuf apart_v, apart_h, apart_x[2], rowtmp;   // per-16-column slot partial sums (ActArgs::rowacc / rowdot_out)
    double *alogw = nullptr;                       // [ais_rows] log-weights, accumulated in double in a fixed order
    DevBuf ais_send, ais_recv;                     // bm_dbm_ais_sharded: this rank's values / the all-gathered values
    // fast-binary mode (bm_bf3.h, bm_dbm_set_fast_binary): bf16 planes of W_l (x = below unit, k = above unit) and of
    // W_l^T, bf16 shadows of the AIS state matrices; `fast_now` is set while a sweep with all-binary states runs
    int ais_literal = 0;                           // bm_dbm_set_ais_literal: float32 accumulation in the reference's order
    int sigmoid_literal = 0;                       // bm_dbm_set_sigmoid_literal: every Bernoulli activation as float32 1 / (1 + exp(-x))
    int fast = 0;
    bool fast_now = false;
    bool fast_ais = false;                         // the running fast sweep is AIS (fp32 copies of v / h2 are not needed)
    Mat16 W3[MAXL], W3t[MAXL];
    Mat16 ax16, ax2_16, av16, ah2_16;
    // PCD particles: one shadow per physical buffer (the Mat structs swap, the buffers keep their shadows)
    Mat16 pv16[2], pH16[MAXL][2];
    const float *pv_key[2] = {nullptr, nullptr}, *pH_key[MAXL][2] = {};
    // a particle shadow is valid once a launch of the running call has written it (the particles a call starts from
    // may be non-binary initial values: they are read in fp32)
    bool pv_ok[2] = {false, false}, pH_ok[MAXL][2] = {};
    uint64_t seed = 0;
    uint32_t call = 0;
    int64_t row0 = 0, prow0 = 0;
};

static void issue_act(bm_dbm *h, const ActArgs &a) { launch_act(a, h->cur); }

static PhiloxKey dkey(const bm_dbm *h, uint32_t site, int t, uint64_t seed, uint32_t call) {
    PhiloxKey k;
    k.k0 = (uint32_t)seed; k.k1 = (uint32_t)(seed >> 32);
    k.site = site + 16u * (uint32_t)t;
    k.call = call;
    return k;
}

// the bf16 shadow of a state matrix of the running fast-binary sweep (null: none)
static const Mat16 *fast_shadow(bm_dbm *h, const float *p, bool **ok = nullptr) {
    if (p == h->ax.p) return &h->ax16;
    if (p == h->ax2.p) return &h->ax2_16;
    if (p == h->av.p) return &h->av16;
    if (p == h->ah2.p) return &h->ah2_16;
    for (int b = 0; b < 2; ++b) {
        if (p && p == h->pv_key[b]) { if (ok) *ok = &h->pv_ok[b]; return &h->pv16[b]; }
        for (int i = 0; i < h->L; ++i) if (p && p == h->pH_key[i][b]) { if (ok) *ok = &h->pH_ok[i][b]; return &h->pH16[i][b]; }
    }
    return nullptr;
}
// the shadow of an INPUT state matrix: only one whose contents are known to mirror the fp32 matrix
static const Mat16 *fast_shadow_in(bm_dbm *h, const float *p) {
    bool *ok = nullptr;
    const Mat16 *m = fast_shadow(h, p, &ok);
    return (m && (!ok || *ok)) ? m : nullptr;
}

// (re)build the bf16 weight planes from the current parameters (fast-binary mode; cheap next to any sweep), on stream
// `st`.  skip_t0: the planes of W_0^T are not needed (the visible layer of the sweep is not a bitmap)
static int fast_build_planes(bm_dbm *h, hipStream_t st = nullptr, bool skip_t0 = false) {
    if (!st) st = h->stream;
    for (int l = 0; l < h->L; ++l) {
        const int a = h->n[l], b = h->n[l + 1];
        if (h->W3[l].rows != a || h->W3[l].cols != b) { BM_TRY(h->W3[l].alloc(3, a, b)); BM_TRY(h->W3t[l].alloc(3, b, a)); }
        hipLaunchKernelGGL(split3_kernel, dim3(1024), dim3(256), 0, st, (const float *)h->W[l].p, h->W[l].ld, a, b,
                           h->W3[l].p, h->W3[l].plane_stride(), h->W3[l].ld, 0);
        if (!(skip_t0 && l == 0))
            hipLaunchKernelGGL(split3_kernel, dim3(1024), dim3(256), 0, st, (const float *)h->W[l].p, h->W[l].ld, a, b,
                               h->W3t[l].p, h->W3t[l].plane_stride(), h->W3t[l].ld, 1);
    }
    BM_HIP(hipGetLastError());
    return 0;
}

// Single-segment passes read their weights x-major ([i][k], k contiguous: one ds_read_b128 per 16 k and lane where the
// k-major image needs four ds_read_b32; 12.95 -> 12.5 us per pass at 784 x 1024, bm_rbm.hip) - the engine keeps W_l and
// W_l^T anyway, so the x-major image of either direction is the OTHER matrix.  BM355_DEBUG=dbm_xm=0: k-major as before.
static bool dbm_xm() {
    static const bool on = !(bm::dbg("dbm_xm") && atoi(bm::dbg("dbm_xm")) == 0);
    return on;
}

// one layer update: out = act(mult * (below.W_lo [+ above.W_hi^T]) + bmult * bias)
//   below [J][n_lo] (pitch ldb) with W_lo = W[lo] ([n_lo][I]);  above [J][n_hi] with Wt[lo+1] ([n_hi][I])
struct LayerIn { const float *p; int ld; };
static void layer_update(bm_dbm *h, int layer /* hidden layer index, -1 = visible */, int J,
                         LayerIn below, LayerIn above, float mult, float bmult, int sample,
                         float *means, float *states, int ldo, const PhiloxKey &key, int64_t row0,
                         const float *prev = nullptr, unsigned *maxdiff = nullptr, ActArgs *extra = nullptr) {
    ActArgs a;
    if (extra) a = *extra; else memset(&a, 0, sizeof(a));
    if (layer >= 0) {
        a.I = h->n[layer + 1];
        a.P1 = make_operand(h->W[layer].p, h->W[layer].ld, a.I);          // W[layer][k = below][i]
        a.Q1 = make_operand(below.p, below.ld, J);
        a.K1 = h->n[layer];
        if (above.p) {
            a.P2 = make_operand(h->Wt[layer + 1].p, h->Wt[layer + 1].ld, a.I);   // W[layer+1]^T [k = above][i]
            a.Q2 = make_operand(above.p, above.ld, J);
            a.K2 = h->n[layer + 2];
        } else if (dbm_xm() && (a.K1 & 3) == 0) {
            a.P1 = make_operand(h->Wt[layer].p, h->Wt[layer].ld, a.I);           // W[layer]^T [i][k = below], x-major
            a.p_xm = 1;
        }
        a.bias = h->hb[layer].p; a.sigma = nullptr; a.kind = BM_UNIT_BERNOULLI;
    } else {
        a.I = h->V;
        a.P1 = make_operand(h->Wt[0].p, h->Wt[0].ld, a.I);               // W[0]^T [k = h0][i = v]
        a.Q1 = make_operand(above.p, above.ld, J);
        a.K1 = h->n[1];
        if (dbm_xm() && (a.K1 & 3) == 0) { a.P1 = make_operand(h->W[0].p, h->W[0].ld, a.I); a.p_xm = 1; }   // W[0] [i = v][k = h0]
        a.bias = h->vb.p; a.sigma = h->sigma.p; a.kind = h->cfg.v_unit;
    }
    if (extra && extra->kind == 2) a.kind = 2;     // raw pre-activation requested (mean-field hoist)
    a.J = J;
    a.mult = mult; a.bmult = bmult;
    a.sample = sample;
    a.means = means; a.states = states; a.ldo = ldo;
    a.key = key; a.row0 = row0;
    a.prev = prev; a.maxdiff = maxdiff;
    a.lit = h->sigmoid_literal;
    if (h->cur == h->stream2 && h->pcd_geo_now) a.geo_hint = h->pcd_geo_now;   // a pass that runs beside the mean-field loop
    if (h->fast_now && !h->multinomial(layer) && a.kind != 2) {
        // fast-binary: the same contraction from the bf16 weight planes and the bf16 shadows of the {0,1} inputs
        // (a state matrix without a valid shadow - real-valued visibles, the first PCD sweep - keeps the fp32 path)
        const Mat16 *sb = below.p ? fast_shadow_in(h, below.p) : nullptr;
        const Mat16 *sa = above.p ? fast_shadow_in(h, above.p) : nullptr;
        auto opnd = [](const Mat16 &m, int nx) { Bf3Operand o; o.p = m.p; o.plane_stride = m.plane_stride(); o.ld = m.ld; o.nx = nx; return o; };
        bool ok = false;
        Bf3Range r;
        memset(&r, 0, sizeof(r));
        if (layer >= 0 && sb && (!above.p || sa)) {
            r.P1 = opnd(h->W3t[layer], a.I); r.Q1 = opnd(*sb, J); r.K1 = sb->ld;          // W_layer^T: [i][k = below]
            if (above.p) { r.P2 = opnd(h->W3[layer + 1], a.I); r.Q2 = opnd(*sa, J); r.K2 = sa->ld; }   // W_{layer+1}: [i][k = above]
            ok = true;
        } else if (layer < 0 && sa) {
            r.P1 = opnd(h->W3[0], a.I); r.Q1 = opnd(*sa, J); r.K1 = sa->ld;               // W_0: [i = v][k = h0]
            ok = true;
        }
        if (ok) a.b3 = r;
        // the shadow of what this launch writes, when that is a sampled bitmap (written by either path's epilogue)
        bool *so_ok = nullptr;
        const Mat16 *so = (states && sample && a.kind == BM_UNIT_BERNOULLI) ? fast_shadow(h, states, &so_ok) : nullptr;
        if (so) {
            a.states16 = so->p; a.ld16 = so->ld;
            if (so_ok) *so_ok = true;
            // AIS: the fp32 copy of a state matrix that only fast-binary launches read is not written at all (visible /
            // top-layer states: 145 MB per beta); x keeps it for the x.hb0 partial sums of the epilogue
            if (ok && h->fast_ais && !a.rowdot_out) a.states = nullptr;
        }
    }
    if (h->multinomial(layer) && a.kind != 2) {
        // MultinomialLayer inside the stack (layers.py:54-70): the GEMM writes the logits mult*z + bmult*b, then one
        // wave per row does the softmax (activation = n_samples * softmax) and, when sampling, the n_samples
        // categorical draws (counts); same two kernels as the MultinomialRBM hidden layer (bm_rbm.hip launch_up)
        float *lg = means; int ldl = ldo;
        if (!lg) {                                 // sampled sweep: the means are not kept, they pass through a row store
            if (h->logit_rows[layer] < J) {
                // (bm_dbm_create preallocates max(N, M) rows: this only grows the store for an unusual row count)
                h->logits[layer].release();
                if (h->logits[layer].alloc(J, a.I)) { h->failed = true; h->logit_rows[layer] = 0; return; }
                h->logit_rows[layer] = J;
            }
            lg = h->logits[layer].p; ldl = h->logits[layer].ld;
        }
        a.kind = 3; a.sample = 0; a.means = lg; a.ldo = ldl; a.states = nullptr; a.negmeans = nullptr;
        a.prev = nullptr; a.maxdiff = nullptr; a.maxdiff_blk = nullptr;
        launch_act(a, h->cur);
        SmArgs m;
        memset(&m, 0, sizeof(m));
        m.L = lg; m.ld = ldl; m.ld_states = ldo; m.I = a.I; m.J = J; m.M = h->cfg.n_samples[layer]; m.sample = sample;
        m.states = (states && (sample || states != lg)) ? states : nullptr;
        m.key = key; m.row0 = row0;
        m.prev = prev; m.ld_prev = ldo; m.maxdiff = maxdiff; m.skip = a.skip;
        hipLaunchKernelGGL(softmax_multinomial_kernel, dim3(J), dim3(64), 2 * (size_t)a.I * sizeof(float), h->cur, m);
        return;
    }
    issue_act(h, a);
}

// `_make_gibbs_step` (dbm.py:385-427): bottom-up Gauss-Seidel sweep.
//   vin / Hin  : current states (Hin[i] may alias mu)      vout / Hout : new states
//   out_means  : Hout receives means (sample == 0) or samples (per-layer flags)
static void gibbs_sweep(bm_dbm *h, int J, LayerIn vin, const Mat *Hin, Mat *vout, Mat *Hout,
                        bool update_v, bool sample, int t, int64_t row0,
                        unsigned *maxdiff = nullptr, const Mat *xw0 = nullptr, const int *skip = nullptr,
                        float *mfblk = nullptr, const float *chk_slots = nullptr) {
    const int L = h->L;
    for (int i = 0; i < L; ++i) {
        LayerIn below = (i == 0) ? vin : LayerIn{Hout[i - 1].p, Hout[i - 1].ld};       // NEW below   :400-402
        LayerIn above = (i + 1 < L) ? LayerIn{Hin[i + 1].p, Hin[i + 1].ld} : LayerIn{nullptr, 0};   // OLD above
        const int smp = sample && h->cfg.sample_h_states[i];
        ActArgs e;
        memset(&e, 0, sizeof(e));
        e.skip = skip;
        e.maxdiff_blk = (maxdiff && mfblk) ? mfblk + (size_t)i * BM_MF_SLOTS : nullptr;
        if (i == 0 && chk_slots) {     // the sweep's first kernel evaluates the loop control of the previous sweep
            e.chk_ctl = h->ctl; e.chk_slots = chk_slots; e.chk_n = L * BM_MF_SLOTS; e.chk_tol = h->cfg.mf_tol;
        }
        if (i == 0 && xw0 && above.p && !h->multinomial(0)) {
            // mean-field: X.W0 is loop invariant — start the chain from the hoisted partial sum and
            // stream only the top-down segment (bit-identical to recomputing X.W0 every sweep)
            e.acc_init = xw0->p; e.ld_init = xw0->ld;
            e.I = h->n[1]; e.J = J;
            e.P1 = make_operand(h->Wt[1].p, h->Wt[1].ld, e.I);
            e.Q1 = make_operand(above.p, above.ld, J);
            e.K1 = h->n[2];
            if (dbm_xm() && (e.K1 & 3) == 0) { e.P1 = make_operand(h->W[1].p, h->W[1].ld, e.I); e.p_xm = 1; }   // W[1] [i = h1][k = h2]
            e.bias = h->hb[0].p; e.kind = BM_UNIT_BERNOULLI;
            e.mult = 1.f; e.bmult = 1.f; e.sample = 0; e.lit = h->sigmoid_literal;
            e.means = Hout[0].p; e.ldo = Hout[0].ld;
            e.prev = maxdiff ? Hin[0].p : nullptr; e.maxdiff = maxdiff;
            issue_act(h, e);
            continue;
        }
        // without sampling the layer's value is its mean: write it as `means` only
        layer_update(h, i, J, below, above, 1.f, 1.f, smp, smp ? nullptr : Hout[i].p, smp ? Hout[i].p : nullptr,
                     Hout[i].ld, dkey(h, SITE_DBM_H + i, t, h->seed, h->call), row0,
                     maxdiff ? Hin[i].p : nullptr, maxdiff, &e);
    }
    if (update_v) {                                                                       // :419-425
        const int smp = sample && h->cfg.sample_v_states;
        layer_update(h, -1, J, LayerIn{nullptr, 0}, LayerIn{Hout[0].p, Hout[0].ld}, 1.f, 1.f, smp,
                     smp ? nullptr : vout->p, smp ? vout->p : nullptr, vout->ld,
                     dkey(h, SITE_DBM_V, t, h->seed, h->call), row0);
    }
}

static int read_flag(bm_dbm *h, float *out) {
    unsigned bits = 0;
    BM_HIP(hipMemcpyAsync(&bits, h->flag, sizeof(bits), hipMemcpyDeviceToHost, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
    memcpy(out, &bits, sizeof(float));
    // data-parallel: the loop condition (dbm.py:449-452) is over ALL rows, i.e. the max over ranks
    if (h->mf_reduce) *out = h->mf_reduce(*out, h->mf_ctx);
    return 0;
}

// `_make_mf` (dbm.py:429-478).  Leaves the result in h->mu; returns executed sweeps.
// The loop trip count is data dependent (residual > tol).  The sweeps are enqueued in groups without host
// round trips — a device-side control word (MfCtl) latches `done` and every later launch returns at once.
// The first group is as long as the previous call's trip count + 1, the host reads the control word of a
// group from a pinned mirror while the next group is already queued.  With the library's communicator
// installed (bm_dbm_set_comm) the residual is all-reduced (max) on the device, in stream order, per sweep;
// only the host-callback hook (bm_dbm_set_mf_allreduce) costs a host round trip per sweep.
// the part of `_make_mf` in front of the loop: approximate-inference init, the hoisted X.W0, the step-0 condition
static bool mf_self_ctl(const bm_dbm *h) {
    // Self-controlled sweeps (single GPU, Bernoulli layers, grids that fit the residual slots), see mean_field()
    bool ok = !h->comm && !h->xchg && !h->mf_reduce;
    for (int i = 0; i < h->L; ++i)
        ok = ok && !h->multinomial(i) && ((h->n[i + 1] + 31) / 32) * ((h->N + 31) / 32) <= BM_MF_SLOTS;
    return ok;
}
static int mf_prologue(bm_dbm *h, const float *X_dev, bool &hoist) {
    const int L = h->L, N = h->N;
    static const bool fold = !(bm::dbg("mf_fold") && atoi(bm::dbg("mf_fold")) == 0);   // =0: the separate residual kernels (A/B)
    // cond at step 0 compares the persistent mu with the init values (:449-452): every init pass leaves max |mu_new - mu| of
    // its tile in its residual slot (or, where there are no slots, in the atomic cell) - the mean-field passes' own epilogue
    // path - instead of two more kernels reading both matrices again
    BM_HIP(hipMemsetAsync(h->flag, 0, sizeof(unsigned), h->stream));
    const bool slots = fold && !h->mf_reduce;                  // (the host-callback path reads the atomic cell alone)
    if (slots && mf_self_ctl(h)) BM_HIP(hipMemsetAsync(h->mfblk.p, 0, 2 * (size_t)MAXL * BM_MF_SLOTS * sizeof(float), h->stream));
    // hoisted loop invariant of the sweeps: z0 = X.W0 (raw chain, no activation)
    hoist = L >= 2 && !h->multinomial(0);
    if (hoist) {
        ActArgs e;
        memset(&e, 0, sizeof(e));
        e.kind = 2;
        layer_update(h, 0, N, LayerIn{X_dev, h->V}, LayerIn{nullptr, 0}, 1.f, 1.f, 0, h->xw0.p, nullptr, h->xw0.ld,
                     dkey(h, 0, 0, h->seed, h->call), 0, nullptr, nullptr, &e);
    }
    // approximate-inference init into the mu_new VARIABLES (:434-446): doubled bottom-up pass
    for (int i = 0; i < L; ++i) {
        LayerIn below = (i == 0) ? LayerIn{X_dev, h->V} : LayerIn{h->mu_new[i - 1].p, h->mu_new[i - 1].ld};
        const float mult = (i == 0 || i < L - 1) ? 2.f : 1.f;
        float *blk = slots ? h->mfblk.p + (size_t)i * BM_MF_SLOTS : nullptr;
        if (i == 0 && hoist && fold) {
            // the first layer's init is the activation of the chain just stored: sigmoid(2 z0 + hb) elementwise, same bits
            const int nwg = N < 256 ? N : 256;
            hipLaunchKernelGGL(mf_init0_kernel, dim3(nwg), dim3(256), 0, h->stream, (const float *)h->xw0.p, h->xw0.ld,
                               (const float *)h->hb[0].p, (const float *)h->mu[0].p, h->mu[0].ld, h->mu_new[0].p, h->mu_new[0].ld,
                               N, h->n[1], mult, 1.f, h->sigmoid_literal ? 1 : 0, h->flag, blk);
            continue;
        }
        ActArgs e;
        memset(&e, 0, sizeof(e));
        e.maxdiff_blk = blk;
        layer_update(h, i, N, below, LayerIn{nullptr, 0}, mult, 1.f, 0, h->mu_new[i].p, nullptr, h->mu_new[i].ld,
                     dkey(h, 0, 0, h->seed, h->call), 0, fold ? h->mu[i].p : nullptr, fold ? h->flag : nullptr, fold ? &e : nullptr);
    }
    if (!fold)
        for (int i = 0; i < L; ++i)
            hipLaunchKernelGGL(maxabsdiff_kernel, dim3(N < 256 ? N : 256), dim3(256), 0, h->stream, (const float *)h->mu[i].p, h->mu[i].ld,
                               (const float *)h->mu_new[i].p, h->mu_new[i].ld, N, h->n[i + 1], h->flag);
    return 0;
}

// `mid` (optional) is called ONCE, after the part in front of the loop is in the queue and before the first sweep: work for a
// second stream (the particle sweeps of a training update) is enqueued there - behind the few launches the critical chain starts
// with, in front of the (up to 2 x max_mf_updates) launches of the loop, so that neither a slow host nor a long loop decides
// when the second stream starts.
struct MfMid { int (*fn)(bm_dbm *, void *); void *ctx; bool called; };
static int mf_mid(bm_dbm *h, MfMid *m) {
    if (!m || m->called) return 0;
    m->called = true;
    return m->fn(h, m->ctx);
}
static int mean_field(bm_dbm *h, const float *X_dev, int *out_n, MfMid *mid = nullptr) {
    const int L = h->L, N = h->N;
    constexpr int MF_GROUP = 8;
    bool hoist = false;
    BM_TRY(mf_prologue(h, X_dev, hoist));
    BM_TRY(mf_mid(h, mid));
    int step = 0;
    Mat *cur = h->mu, *alt = h->mu_alt;
    if (h->mf_reduce) {
        float diff = 0.f;
        BM_TRY(read_flag(h, &diff));
        // body (:454-457): mu_new = sweep(X, mu) (values, not the mu_new variables), then swap
        while (step < h->cfg.max_mf_updates && diff > h->cfg.mf_tol) {
            BM_HIP(hipMemsetAsync(h->flag, 0, sizeof(unsigned), h->stream));
            gibbs_sweep(h, N, LayerIn{X_dev, h->V}, cur, nullptr, alt, false, false, 0, 0, h->flag,
                        hoist ? &h->xw0 : nullptr);
            BM_TRY(read_flag(h, &diff));
            Mat *t = cur; cur = alt; alt = t;
            ++step;
        }
    } else {
        // loop control of one sweep: local (one kernel) or global (residual all-reduced over the ranks in stream
        // order: every rank enqueues the same sequence and latches the same `done`, so no host round trip is
        // needed and the ranks stay in lockstep)
        auto ctl_step = [&](int init) -> int {
            if (!h->comm && !h->xchg) {
                hipLaunchKernelGGL(mf_ctl_kernel, dim3(1), dim3(256), 0, h->stream, h->ctl, h->cfg.mf_tol, init,
                                   h->mfblk.p, h->L * BM_MF_SLOTS);
                return 0;
            }
            if (h->xchg)     // residual -> max over the ranks -> latch in ONE launch (three launches per sweep were 0.4 ms of a
                             // 1.8 ms data-parallel update at 45 sweeps)
                return xchg_mf_ctl_step(h->xchg, h->ctl, h->mfblk.p, h->L * BM_MF_SLOTS, h->cfg.mf_tol, init, h->stream);
            hipLaunchKernelGGL(mf_resid_kernel, dim3(1), dim3(256), 0, h->stream, h->ctl, h->mfblk.p, h->L * BM_MF_SLOTS);
            BM_TRY(bm_comm_allreduce_max(h->comm, &h->ctl->resid, 1, (void *)h->stream));
            hipLaunchKernelGGL(mf_latch_kernel, dim3(1), dim3(64), 0, h->stream, h->ctl, h->cfg.mf_tol, init);
            return 0;
        };
        // Self-controlled sweeps (single GPU, Bernoulli layers, grids that fit the residual slots): the loop-control
        // update of sweep s-1 is evaluated by the first kernel of sweep s (ActArgs::chk_ctl) from the slot set of
        // the other parity; only the LAST sweep of a group needs the one-workgroup control kernel.  Otherwise
        // (communicator installed, Multinomial layers, > BM_MF_SLOTS workgroups possible) one control step per sweep.
        const bool self_ctl = mf_self_ctl(h);
        const size_t set_sz = (size_t)MAXL * BM_MF_SLOTS;
        {
            static const bool fold = !(bm::dbg("mf_fold") && atoi(bm::dbg("mf_fold")) == 0);
            if (self_ctl && !fold) BM_HIP(hipMemsetAsync(h->mfblk.p, 0, 2 * set_sz * sizeof(float), h->stream));   // (else: mf_prologue)
        }
        BM_TRY(ctl_step(1));
        // Groups of sweeps are enqueued without host round trips; the loop-control record is copied to a pinned
        // mirror after each group and READ ONE GROUP LATE: group g+1 is already in the queue when the host looks at
        // group g, so the GPU never idles waiting for the host (sweeps enqueued past the end of the loop return at
        // once).  The first group is as long as the previous call's trip count - for a training run the count
        // barely moves from one minibatch to the next - so a typical call costs one or two reads.
        constexpr int R = bm_dbm::MF_RING;
        int enq = 0, g_enq = 0, g_read = 0;
        MfCtl host;
        host.done = 0; host.steps = 0;
        const int max_it = h->cfg.max_mf_updates;
        auto enqueue_group = [&](int g) -> int {
            for (int s = 0; s < g; ++s) {
                // sweep number enq+s runs only if all before it ran, so its ping-pong parity is static
                const int sw = enq + s;
                Mat *src = (sw & 1) ? h->mu_alt : h->mu, *dst = (sw & 1) ? h->mu : h->mu_alt;
                if (self_ctl) {
                    float *mine = h->mfblk.p + (size_t)(sw & 1) * set_sz, *prev = h->mfblk.p + (size_t)((sw & 1) ^ 1) * set_sz;
                    gibbs_sweep(h, N, LayerIn{X_dev, h->V}, src, nullptr, dst, false, false, 0, 0, h->flag,
                                hoist ? &h->xw0 : nullptr, &h->ctl->done, mine, s > 0 ? prev : nullptr);
                    if (s == g - 1)      // Check(last sweep of the group); it also clears the slots it read
                        hipLaunchKernelGGL(mf_ctl_kernel, dim3(1), dim3(256), 0, h->stream, h->ctl, h->cfg.mf_tol, 0,
                                           mine, h->L * BM_MF_SLOTS);
                } else {
                    gibbs_sweep(h, N, LayerIn{X_dev, h->V}, src, nullptr, dst, false, false, 0, 0, h->flag,
                                hoist ? &h->xw0 : nullptr, &h->ctl->done, h->mfblk.p);
                    BM_TRY(ctl_step(0));
                }
            }
            enq += g;
            BM_HIP(hipMemcpyAsync(&h->ctl_host[g_enq % R], h->ctl, sizeof(MfCtl), hipMemcpyDeviceToHost, h->stream));
            BM_HIP(hipEventRecord(h->ctl_ev[g_enq % R], h->stream));
            ++g_enq;
            return 0;
        };
        if (max_it <= 0) {                 // no sweeps: fetch the step-0 record
            BM_HIP(hipMemcpyAsync(&h->ctl_host[0], h->ctl, sizeof(MfCtl), hipMemcpyDeviceToHost, h->stream));
            BM_HIP(hipStreamSynchronize(h->stream));
            host = h->ctl_host[0];
        }
        auto read_oldest = [&]() -> int {
            BM_HIP(hipEventSynchronize(h->ctl_ev[g_read % R]));
            host = h->ctl_host[g_read % R];
            ++g_read;
            return 0;
        };
        if (max_it > 0) {
            // first group: the previous trip count + 1 (a sweep too many costs two kernels that return at once,
            // a sweep too few costs a host round trip), read before anything else is enqueued
            int g0 = h->mf_pred > 0 ? h->mf_pred + 1 : MF_GROUP;
            if (g0 > max_it) g0 = max_it;
            BM_TRY(enqueue_group(g0));
            BM_TRY(read_oldest());
            // not converged yet: short groups, one of them always queued behind the one being read
            while (!host.done && (enq < max_it || g_read < g_enq)) {
                while (enq < max_it && g_enq - g_read < 2)
                    BM_TRY(enqueue_group(max_it - enq < MF_GROUP / 2 ? max_it - enq : MF_GROUP / 2));
                BM_TRY(read_oldest());
            }
            if (g_read < g_enq) {          // groups enqueued past the end of the loop do nothing; free their ring slots
                BM_HIP(hipEventSynchronize(h->ctl_ev[(g_enq - 1) % R]));
                g_read = g_enq;
            }
        }
        h->mf_pred = host.steps > 0 ? host.steps : 0;
        step = host.steps;
        if (step & 1) { cur = h->mu_alt; alt = h->mu; }
    }
    if (cur != h->mu)                  // `self._mu[i].assign(mu[i])` (:477): keep the handle's mu as the result
        for (int i = 0; i < L; ++i) { Mat t = h->mu[i]; h->mu[i] = h->mu_alt[i]; h->mu_alt[i] = t; }
    if (out_n) *out_n = step;
    return 0;
}

// `_make_particles_update` (dbm.py:480-509)
// fast-binary PCD: the hidden layers are sampled Bernoulli layers (bitmaps from the second sweep on), the visible one
// may be real valued (then only the top-down half of each sweep runs on the bf16 cores)
static bool fast_pcd_ok(const bm_dbm *h, bool sample) {
    // (level 1: only where the particle sweeps gain - from 8M weights in the bottom layer upwards; at 784-512-1024 they lose,
    //  profiles/r5_dbm_summary.md.  Level 2: wherever legal)
    if (!h->fast || !sample) return false;
    if (h->fast < 2 && (long long)h->V * h->n[1] < (8ll << 20)) return false;
    for (int i = 0; i < h->L; ++i) if (h->multinomial(i) || !h->cfg.sample_h_states[i]) return false;
    return true;
}
static int fast_pcd_begin(bm_dbm *h) {
    const bool vbits = h->cfg.v_unit == BM_UNIT_BERNOULLI && h->cfg.sample_v_states;
    if (!h->pH_key[0][0]) {                       // one shadow per physical particle buffer
        Mat *vb[2] = {&h->v, &h->v_new};
        for (int b = 0; b < 2; ++b) {
            if (vbits) { BM_TRY(h->pv16[b].alloc(1, h->M, h->V)); h->pv_key[b] = vb[b]->p; }
            for (int i = 0; i < h->L; ++i) {
                Mat *hb = b ? &h->H_new[i] : &h->H[i];
                BM_TRY(h->pH16[i][b].alloc(1, h->M, h->n[i + 1])); h->pH_key[i][b] = hb->p;
            }
        }
    }
    for (int b = 0; b < 2; ++b) { h->pv_ok[b] = false; for (int i = 0; i < h->L; ++i) h->pH_ok[i][b] = false; }
    return fast_build_planes(h, h->cur, !vbits);  // the parameters changed since the last update
}

static void particles_update(bm_dbm *h, int k, bool sample, bool update_only_v_at_end = false) {
    (void)update_only_v_at_end;
    struct FastScope { bm_dbm *h; bool on; ~FastScope() { if (on) h->fast_now = false; } } scope{h, false};
    if (fast_pcd_ok(h, sample)) {
        if (fast_pcd_begin(h) == 0) { scope.on = true; h->fast_now = true; h->fast_ais = false; }
        else h->failed = true;
    }
    for (int t = 0; t < k; ++t) {
        // (fast-binary: sweep 0 reads the particles it starts from in fp32 and leaves shadows of what it samples; from
        // then on every sampled Bernoulli input is a bitmap with a valid shadow)
        gibbs_sweep(h, h->M, LayerIn{h->v.p, h->v.ld}, h->H, &h->v_new, h->H_new, true, sample, t, h->prow0);
        Mat tv = h->v; h->v = h->v_new; h->v_new = tv;                    // swap particles (:493)
        for (int i = 0; i < h->L; ++i) { Mat th = h->H[i]; h->H[i] = h->H_new[i]; h->H_new[i] = th; }
    }
}


// mean-field on the data rows and PCD sweeps on the particles of one update, concurrently (see bm_dbm::stream2).
// The particle sweeps go to the second stream, between a fork event (recorded before anything of this update is enqueued: the
// previous parameter update) and a join event the main stream waits for before anything reads the particles; the host enqueues
// them from mean_field()'s `mid` hook - after the launches in front of the mean-field loop, before the loop's own.  Same kernels,
// same RNG streams: results do not change.
static bool pcd_overlap_ok(const bm_dbm *h) {
    static const bool off = bm::dbg("dbm_overlap") && atoi(bm::dbg("dbm_overlap")) == 0;
    if (off || h->updates_seen < 2) return false;          // the first updates tune their launches undisturbed
    for (int i = 0; i < h->L; ++i) if (h->multinomial(i)) return false;     // one logits row store per layer
    return true;
}
static int enqueue_particles_stream2(bm_dbm *h, void *ctx) {
    const int k = *(const int *)ctx;
    BM_HIP(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
    static const int min_ratio = bm::dbg("dbm_pcd_ratio") ? atoi(bm::dbg("dbm_pcd_ratio")) : 1;
    h->pcd_geo_now = (h->pcd_geo && h->mf_pred > min_ratio * k) ? h->pcd_geo : 0;
    h->cur = h->stream2;
    particles_update(h, k, true);                                 // :521
    h->cur = h->stream;
    h->pcd_geo_now = 0;
    BM_HIP(hipEventRecord(h->ev_join, h->stream2));
    return 0;
}
static int mean_field_and_particles(bm_dbm *h, const float *X_dev, int k, int *out_n) {
    const bool ov = pcd_overlap_ok(h);
    // where the host enqueues the particle sweeps: behind the mean-field's prologue (default) or in front of it (=0); same time
    // per update on a fast host (profiles/r6_dbm_ab.txt)
    static const bool late = !(bm::dbg("dbm_pcd_late") && atoi(bm::dbg("dbm_pcd_late")) == 0);
    MfMid mid{enqueue_particles_stream2, &k, false};
    if (ov) {
        BM_HIP(hipEventRecord(h->ev_fork, h->stream));            // = the previous parameter update
        if (!late) BM_TRY(mf_mid(h, &mid));
    }
    int rc = mean_field(h, X_dev, out_n, ov ? &mid : nullptr);    // :517
    if (ov && !mid.called) {                                      // the mean-field failed before its hook ran
        if (mf_mid(h, &mid) && !rc) rc = 1;
    }
    // (the join is enqueued even when the mean-field failed: later calls on the main stream must not race the
    //  particle sweeps still running on the second one)
    if (ov) { if (hipStreamWaitEvent(h->stream, h->ev_join, 0) != hipSuccess && !rc) { set_error("hipStreamWaitEvent(join) failed"); return 1; } }
    else if (!rc) particles_update(h, k, true);
    h->updates_seen++;
    return rc;
}

static void xchg_dw_replaced(bm_xchg *x);
static int check_dw(const bm_dbm *h, const char *what) {
    BM_CHECK(!h->dw_sharded, "%s: after bm_dbm_exchange_apply_direct this rank holds only its column slices of the momentum "
                             "buffers dW; call bm_dbm_exchange_gather_dw (DirectExchange.gather_dw) on every rank first", what);
    return 0;
}

static size_t sums_off(const bm_dbm *h, int which /* 0: X, 1: v, 2+2i: mu_i, 3+2i: H_i */) {
    size_t o = 0;
    if (which == 0) return 0;
    o += h->V;
    if (which == 1) return o;
    o += h->V;
    for (int i = 0; i < h->L; ++i) {
        if (which == 2 + 2 * i) return o;
        o += h->n[i + 1];
        if (which == 3 + 2 * i) return o;
        o += h->n[i + 1];
    }
    return o;
}

// raw column sums of X, v, mu_i, H_i (dbm.py:553, :573-576, :581-586)
static void launch_dbm_colsums(bm_dbm *h, const float *X_dev) {
    const int L = h->L;
    ColSumArgs c;
    memset(&c, 0, sizeof(c));
    int nj = 0;
    c.job[nj++] = ColSumJob{X_dev, nullptr, h->V, 0, h->V, h->N, h->sums_p + sums_off(h, 0)};
    c.job[nj++] = ColSumJob{h->v.p, nullptr, h->v.ld, 0, h->V, h->M, h->sums_p + sums_off(h, 1)};
    for (int i = 0; i < L; ++i) {
        c.job[nj++] = ColSumJob{h->mu[i].p, nullptr, h->mu[i].ld, 0, h->n[i + 1], h->N, h->sums_p + sums_off(h, 2 + 2 * i)};
        c.job[nj++] = ColSumJob{h->H[i].p, nullptr, h->H[i].ld, 0, h->n[i + 1], h->M, h->sums_p + sums_off(h, 3 + 2 * i)};
    }
    c.njobs = nj;
    c.first_wave[0] = 0;
    for (int j = 0; j < nj; ++j) c.first_wave[j + 1] = c.first_wave[j] + (c.job[j].ncols + 63) / 64;
    hipLaunchKernelGGL(colsum_kernel, dim3(c.first_wave[nj]), dim3(NT), 0, h->stream, c);
}

// bias / running-mean / sparsity updates from the (possibly all-reduced) column sums
static void launch_dbm_biases(bm_dbm *h, float N, float M, float lr, float mom) {
    static_assert(DBM_BIAS_JOBS == 1 + MAXL, "one job per layer + the visible one");
    DbmBiasMulti m;
    memset(&m, 0, sizeof(m));
    int nmax = h->V;
    {
        DbmBiasArgs &b = m.job[0];
        b.s_pos = h->sums_p + sums_off(h, 0); b.s_neg = h->sums_p + sums_off(h, 1);
        b.b = h->vb.p; b.db = h->dvb.p; b.n = h->V; b.N = N; b.M = M; b.lr = lr; b.mom = mom;
    }
    for (int i = 0; i < h->L; ++i) {
        DbmBiasArgs &b = m.job[1 + i];
        b.s_pos = h->sums_p + sums_off(h, 2 + 2 * i); b.s_neg = h->sums_p + sums_off(h, 3 + 2 * i);
        b.b = h->hb[i].p; b.db = h->dhb[i].p; b.q = h->q[i].p; b.mm = h->mm[i].p; b.pen = h->pen[i].p;
        b.n = h->n[i + 1]; b.layer = i;
        b.N = N; b.M = M; b.lr = lr; b.mom = mom;
        b.damping = h->cfg.sparsity_damping; b.cost = h->cfg.sparsity_cost[i]; b.target = h->cfg.sparsity_target[i];
        if (b.n > nmax) nmax = b.n;
    }
    hipLaunchKernelGGL(dbm_bias_multi_kernel, dim3((nmax + 255) / 256, 1 + h->L), dim3(256), 0, h->stream, m);
}

// outer products of layer i: fused (update in the epilogue) or raw pos / neg into `grad`
static void launch_dbm_grad(bm_dbm *h, const float *X_dev, int i, int fused, float N, float M, float lr, float mom, hipStream_t st = nullptr) {
    GradArgs g;
    memset(&g, 0, sizeof(g));
    const float *below_pos = (i == 0) ? X_dev : h->mu[i - 1].p;
    const int ld_bp = (i == 0) ? h->V : h->mu[i - 1].ld;
    const Mat &below_neg = (i == 0) ? h->v : h->H[i - 1];
    g.Ppos = make_operand(h->mu[i].p, h->mu[i].ld, h->n[i + 1]);      // mu_i         [k = b][i]
    g.Qpos = make_operand(below_pos, ld_bp, h->n[i]);                  // X / mu_{i-1} [k = b][j]
    g.Kpos = h->N;
    g.Pneg = make_operand(h->H[i].p, h->H[i].ld, h->n[i + 1]);
    g.Qneg = make_operand(below_neg.p, below_neg.ld, h->n[i]);
    g.Kneg = h->M;
    g.I = h->n[i + 1]; g.J = h->n[i];
    g.form = 1; g.fused = fused;
    g.raw = h->grad.p + h->raw_off[i][0]; g.raw2 = h->grad.p + h->raw_off[i][1];
    g.W = h->W[i].p; g.dW = h->dW[i].p; g.Wt = nullptr;                // Wt is rewritten by the max-norm pass
    g.ldw = h->W[i].ld; g.ldwt = h->Wt[i].ld;
    g.pen = h->pen[i].p;
    g.N = N; g.M = M; g.l2 = h->cfg.l2; g.lr = lr; g.mom = mom;
    launch_grad(g, st ? st : h->stream);
}

static int recon_msre(bm_dbm *h, const float *X_dev, float *out_msre);

static void launch_dbm_maxnorm(bm_dbm *h, int i, int c_first = 0, int c_end = -1, hipStream_t st = nullptr) {
    if (!st) st = h->stream;
    MaxNormArgs m;
    m.c_first = c_first; m.c_end = c_end < 0 ? h->n[i + 1] : c_end;
    if (m.c_end <= m.c_first) return;
    m.W = h->W[i].p; m.Wt = h->Wt[i].p; m.I = h->n[i + 1]; m.J = h->n[i]; m.ldw = h->W[i].ld; m.ldwt = h->Wt[i].ld;
    m.max_norm = h->cfg.max_norm; m.norm_out = h->wnorm[i].p;
    m.num = h->mn_fac[i].p; m.den = h->mn_fac[i].p + m.I;
    const int nc = m.c_end - m.c_first;
    hipLaunchKernelGGL(maxnorm_kernel, dim3((nc + MN_COLS - 1) / MN_COLS), dim3(NT), 0, st, m);
    hipLaunchKernelGGL(maxnorm_scale_kernel, dim3(((nc + 31) / 32) * ((m.J + 31) / 32)), dim3(256), 0, st, m);
}

// gradients + sparsity + momentum + max-norm (dbm.py:550-621) from the current mu / particles
static int apply_update(bm_dbm *h, const float *X_dev, float lr, float mom) {
    const float N = (float)h->N, M = (float)h->M;
    launch_dbm_colsums(h, X_dev);
    launch_dbm_biases(h, N, M, lr, mom);
    // The layers' outer products + max-norm passes are independent of each other (each reads mu / particles and its own
    // penalty vector, writes its own W / dW / W^T): odd layers go to the second stream, between a fork and a join event, so
    // that the 104 + 128 tiles of a 784-512-1024 stack share the chip instead of taking turns (once the launches are tuned).
    static const bool split_off = bm::dbg("dbm_tail_split") && atoi(bm::dbg("dbm_tail_split")) == 0;
    const bool split = !split_off && h->L >= 2 && h->updates_seen >= 3;
    if (split) {
        BM_HIP(hipEventRecord(h->ev_fork, h->stream));
        BM_HIP(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
    }
    for (int i = 0; i < h->L; ++i) {
        hipStream_t st = (split && (i & 1)) ? h->stream2 : h->stream;
        launch_dbm_grad(h, X_dev, i, 1, N, M, lr, mom, st);
        launch_dbm_maxnorm(h, i, 0, -1, st);
    }
    if (split) {
        BM_HIP(hipEventRecord(h->ev_join, h->stream2));
        BM_HIP(hipStreamWaitEvent(h->stream, h->ev_join, 0));
    }
    BM_CHECK(!h->failed, "bm_dbm: a device allocation failed inside a sweep (row store of a Multinomial layer)");
    BM_HIP(hipGetLastError());
    return 0;
}

// reconstruction sigma(mu0 W0^T + vb) (dbm.py:625-628) into R (pitch ldr)
static void reconstruct_from_mu(bm_dbm *h, float *R, int ldr) {
    layer_update(h, -1, h->N, LayerIn{nullptr, 0}, LayerIn{h->mu[0].p, h->mu[0].ld}, 1.f, 1.f, 0, R, nullptr, ldr,
                 dkey(h, 0, 0, h->seed, h->call), 0);
}

extern "C" {

int bm_dbm_create(const bm_dbm_config *cfg, bm_dbm **out) {
    BM_CHECK(cfg && out, "null argument");
    BM_CHECK(cfg->n_layers >= 1 && cfg->n_layers <= MAXL, "n_layers %d outside [1, %d]", cfg->n_layers, MAXL);
    BM_CHECK(cfg->n_visible >= 1 && cfg->n_particles >= 1 && cfg->batch_size >= 1, "bad sizes");
    BM_CHECK(bm_device_count() > 0, "no HIP device visible: libbm355 has no CPU fallback");
    bm_dbm *h = new bm_dbm();
    h->cfg = *cfg;
    h->L = cfg->n_layers; h->V = cfg->n_visible; h->N = cfg->batch_size; h->M = cfg->n_particles;
    h->n[0] = h->V;
    for (int i = 0; i < h->L; ++i) {
        BM_CHECK(cfg->n_hiddens[i] >= 1, "bad hidden size");
        BM_CHECK(cfg->n_hiddens[i] > i, "layer %d needs more than %d units (sparsity index, dbm.py:583)", i, i);
        BM_CHECK(cfg->h_unit[i] == BM_UNIT_BERNOULLI || cfg->h_unit[i] == BM_UNIT_MULTINOMIAL, "unknown unit %d of hidden layer %d",
                 cfg->h_unit[i], i);
        if (cfg->h_unit[i] == BM_UNIT_MULTINOMIAL) {
            BM_CHECK(cfg->n_samples[i] >= 1, "Multinomial layer %d: n_samples must be >= 1 (got %d)", i, cfg->n_samples[i]);
            BM_CHECK(cfg->n_hiddens[i] <= 8192, "Multinomial layer %d: %d units > 8192 (softmax row staged in LDS)", i, cfg->n_hiddens[i]);
        }
        h->n[i + 1] = cfg->n_hiddens[i];
    }
    BM_HIP(hipStreamCreate(&h->stream));
    BM_HIP(hipStreamCreate(&h->stream2));
    {
        bool small = h->L >= 2 && h->N <= 1024 && h->M <= 1024 && cfg->max_mf_updates >= 2;
        for (int i = 0; i < h->L; ++i) small = small && (long long)h->n[i] * h->n[i + 1] <= (2ll << 20);
        h->pcd_geo = small ? 3 : 0;
        if (bm::dbg("dbm_pcd_geo")) h->pcd_geo = atoi(bm::dbg("dbm_pcd_geo"));
    }
    h->cur = h->stream;
    BM_HIP(hipEventCreate(&h->ev0));
    BM_HIP(hipEventCreate(&h->ev1));
    BM_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    BM_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    BM_HIP(hipHostMalloc((void **)&h->ctl_host, bm_dbm::MF_RING * sizeof(MfCtl)));
    for (int i = 0; i < bm_dbm::MF_RING; ++i) BM_HIP(hipEventCreateWithFlags(&h->ctl_ev[i], hipEventDisableTiming));
    size_t nsums = 2 * (size_t)h->V;
    for (int i = 0; i < h->L; ++i) {
        const int a = h->n[i], b = h->n[i + 1];
        BM_TRY(h->W[i].alloc(a, b)); BM_TRY(h->Wt[i].alloc(b, a)); BM_TRY(h->dW[i].alloc(a, b));
        BM_TRY(h->hb[i].alloc(b)); BM_TRY(h->dhb[i].alloc(b)); BM_TRY(h->q[i].alloc(b)); BM_TRY(h->mm[i].alloc(b));
        BM_TRY(h->pen[i].alloc(b)); BM_TRY(h->wnorm[i].alloc(b)); BM_TRY(h->mn_fac[i].alloc(2 * (size_t)b));
        BM_TRY(h->mu[i].alloc(h->N, b)); BM_TRY(h->mu_alt[i].alloc(h->N, b)); BM_TRY(h->mu_new[i].alloc(h->N, b));
        BM_TRY(h->H[i].alloc(h->M, b)); BM_TRY(h->H_new[i].alloc(h->M, b));
        if (h->multinomial(i)) {                   // row store of the logits: the sweeps cannot fail on an allocation
            const int rows = h->N > h->M ? h->N : h->M;
            BM_TRY(h->logits[i].alloc(rows, b));
            h->logit_rows[i] = rows;
        }
        nsums += 2 * (size_t)b;
    }
    BM_TRY(h->vb.alloc(h->V)); BM_TRY(h->dvb.alloc(h->V)); BM_TRY(h->sigma.alloc(h->V));
    BM_TRY(h->v.alloc(h->M, h->V)); BM_TRY(h->v_new.alloc(h->M, h->V));
    BM_TRY(h->recon.alloc(h->N, h->V));
    {   // one contiguous buffer so that data-parallel training needs ONE all-reduce
        size_t off = 0;
        for (int i = 0; i < h->L; ++i)
            for (int s = 0; s < 2; ++s) { h->raw_off[i][s] = off; off += h->W[i].count(); }
        BM_TRY(h->grad.alloc(off + nsums));
        h->sums_p = h->grad.p + off;
    }
    BM_TRY(h->mfblk.alloc(2 * (size_t)BM_DBM_MAX_LAYERS * BM_MF_SLOTS));
    BM_HIP(hipMalloc((void **)&h->ctl, sizeof(MfCtl)));
    BM_HIP(hipMemset(h->ctl, 0, sizeof(MfCtl)));
    h->flag = &h->ctl->maxdiff;
    BM_TRY(h->xw0.alloc(h->N, h->n[1]));
    BM_HIP(hipMalloc((void **)&h->scal, 4 * sizeof(double)));
    {
        std::vector<float> ones(h->V, 1.0f);
        BM_HIP(hipMemcpy(h->sigma.p, ones.data(), h->V * sizeof(float), hipMemcpyHostToDevice));
    }
    *out = h;
    return 0;
}

int bm_dbm_destroy(bm_dbm *h) {
    if (!h) return 0;
    (void)hipStreamSynchronize(h->stream);
    if (h->xchg_used) xchg_bind_user(h->xchg_used, nullptr);
    for (int i = 0; i < h->L; ++i) {
        Mat *ms[] = {&h->W[i], &h->Wt[i], &h->dW[i], &h->mu[i], &h->mu_alt[i], &h->mu_new[i], &h->H[i], &h->H_new[i]};
        for (Mat *m : ms) m->release();
        DevBuf *bs[] = {&h->hb[i], &h->dhb[i], &h->q[i], &h->mm[i], &h->pen[i], &h->wnorm[i], &h->mn_fac[i]};
        for (DevBuf *b : bs) b->release();
        h->logits[i].release();
        h->W3[i].release(); h->W3t[i].release();
    }
    h->ax16.release(); h->ax2_16.release(); h->av16.release(); h->ah2_16.release();
    for (int b = 0; b < 2; ++b) { h->pv16[b].release(); for (int i = 0; i < MAXL; ++i) h->pH16[i][b].release(); }
    Mat *ms[] = {&h->v, &h->v_new, &h->recon, &h->ax, &h->ax2, &h->av, &h->ah2};
    for (Mat *m : ms) m->release();
    DevBuf *bs[] = {&h->vb, &h->dvb, &h->sigma, &h->grad, &h->apart_v, &h->apart_h, &h->apart_x[0], &h->apart_x[1], &h->rowtmp,
                     &h->ais_send, &h->ais_recv};
    for (DevBuf *b : bs) b->release();
    if (h->alogw) (void)hipFree(h->alogw);
    if (h->ctl) (void)hipFree(h->ctl);
    if (h->ctl_host) (void)hipHostFree(h->ctl_host);
    for (int i = 0; i < bm_dbm::MF_RING; ++i) if (h->ctl_ev[i]) (void)hipEventDestroy(h->ctl_ev[i]);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    h->mfblk.release();
    h->xw0.release();
    if (h->scal) (void)hipFree(h->scal);
    (void)hipEventDestroy(h->ev0);
    (void)hipEventDestroy(h->ev1);
    (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

int bm_dbm_sync(bm_dbm *h) {
    BM_HIP(hipStreamSynchronize(h->stream));
    if (h->xchg_used) BM_TRY(xchg_check_status(h->xchg_used));      // a lost rank is an ERROR here, never a silent wrong sum
    return 0;
}
int bm_dbm_seed(bm_dbm *h, uint64_t seed) { h->seed = seed; h->call = 0; return 0; }
int bm_dbm_set_row_offset(bm_dbm *h, int64_t row0, int64_t particle0) { h->row0 = row0; h->prow0 = particle0; return 0; }

// "W", "W_1", "hb_2" ... -> (base, layer)
static bool split_name(const std::string &nm, std::string &base, int &idx) {
    const size_t u = nm.rfind('_');
    base = nm; idx = 0;
    if (u != std::string::npos && u + 1 < nm.size() && isdigit((unsigned char)nm[u + 1])) {
        bool digits = true;
        for (size_t i = u + 1; i < nm.size(); ++i) digits = digits && isdigit((unsigned char)nm[i]);
        if (digits) { base = nm.substr(0, u); idx = atoi(nm.c_str() + u + 1); }
    }
    return true;
}

static int resolve(bm_dbm *h, const char *name, Mat **mat, DevBuf **vec, bool *is_W) {
    std::string base; int idx;
    split_name(std::string(name ? name : ""), base, idx);
    *mat = nullptr; *vec = nullptr; *is_W = false;
    BM_CHECK(idx >= 0 && idx < h->L, "layer index %d out of range in '%s'", idx, name ? name : "(null)");
    if (base == "W") { *mat = &h->W[idx]; *is_W = true; }
    else if (base == "dW") *mat = &h->dW[idx];
    else if (base == "mu") *mat = &h->mu[idx];
    else if (base == "mu_new") *mat = &h->mu_new[idx];
    else if (base == "h") *mat = &h->H[idx];
    else if (base == "h_new") *mat = &h->H_new[idx];
    else if (base == "v" && idx == 0) *mat = &h->v;
    else if (base == "v_new" && idx == 0) *mat = &h->v_new;
    else if (base == "hb") *vec = &h->hb[idx];
    else if (base == "dhb") *vec = &h->dhb[idx];
    else if (base == "q_means") *vec = &h->q[idx];
    else if (base == "mu_means") *vec = &h->mm[idx];
    else if (base == "W_norm") *vec = &h->wnorm[idx];
    else if (base == "vb" && idx == 0) *vec = &h->vb;
    else if (base == "dvb" && idx == 0) *vec = &h->dvb;
    else if (base == "sigma" && idx == 0) *vec = &h->sigma;
    BM_CHECK(*mat || *vec, "unknown DBM variable '%s'", name ? name : "(null)");
    return 0;
}

int bm_dbm_set_param(bm_dbm *h, const char *name, const float *host, size_t n) {
    Mat *m; DevBuf *v; bool isW;
    BM_TRY(resolve(h, name, &m, &v, &isW));
    BM_HIP(hipStreamSynchronize(h->stream));
    if (m) {
        BM_CHECK(n == (size_t)m->rows * m->cols, "variable '%s' has %zu elements, got %zu", name, (size_t)m->rows * m->cols, n);
        BM_TRY(m->upload(host));
        if (m >= h->dW && m < h->dW + MAXL && h->dw_sharded) {
            // the host replaces a momentum buffer whole: with every layer's buffer replaced the replicas are complete again
            h->dw_set_mask |= 1u << (unsigned)(m - h->dW);
            if (h->dw_set_mask == (1u << (unsigned)h->L) - 1u) { h->dw_sharded = false; if (h->xchg_used) xchg_dw_replaced(h->xchg_used); }
        }
        if (isW) {
            std::vector<float> t(n);
            for (int r = 0; r < m->rows; ++r)
                for (int c = 0; c < m->cols; ++c) t[(size_t)c * m->rows + r] = host[(size_t)r * m->cols + c];
            Mat *wt = &h->Wt[m - h->W];
            BM_TRY(wt->upload(t.data()));
        }
        return 0;
    }
    BM_CHECK(n == v->n, "variable '%s' has %zu elements, got %zu", name, v->n, n);
    BM_HIP(hipMemcpy(v->p, host, n * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int bm_dbm_get_param(bm_dbm *h, const char *name, float *host, size_t n) {
    Mat *m; DevBuf *v; bool isW;
    BM_TRY(resolve(h, name, &m, &v, &isW));
    BM_HIP(hipStreamSynchronize(h->stream));
    if (m) {
        BM_CHECK(n == (size_t)m->rows * m->cols, "variable '%s' has %zu elements, got %zu", name, (size_t)m->rows * m->cols, n);
        if (m >= h->dW && m < h->dW + MAXL) BM_TRY(check_dw(h, "bm_dbm_get_param(dW)"));
        return m->download(host);
    }
    BM_CHECK(n == v->n, "variable '%s' has %zu elements, got %zu", name, v->n, n);
    BM_HIP(hipMemcpy(host, v->p, n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int bm_dbm_dev_ptr(bm_dbm *h, const char *name, void **out_dev, size_t *out_n) {
    if (name && std::string(name) == "grad") {
        *out_dev = h->grad.p;
        if (out_n) *out_n = h->grad.n;
        return 0;
    }
    Mat *m; DevBuf *v; bool isW;
    BM_TRY(resolve(h, name, &m, &v, &isW));
    BM_CHECK(v, "no device view for '%s' (matrices are pitched; use get/set_param)", name);
    *out_dev = v->p;
    if (out_n) *out_n = v->n;
    return 0;
}

int bm_dbm_train_step(bm_dbm *h, const float *X_dev, float lr, float mom, int32_t k,
                      int32_t *out_n_mf, float *out_msre) {
    BM_CHECK(k >= 1, "n_gibbs_steps must be >= 1 (got %d)", k);
    BM_TRY(check_dw(h, "bm_dbm_train_step"));
    int nmf = 0;
    BM_TRY(mean_field_and_particles(h, X_dev, k, &nmf));      // :517, :521
    if (out_msre) BM_TRY(recon_msre(h, X_dev, out_msre));     // :625-630 (W before the update)
    BM_TRY(apply_update(h, X_dev, lr, mom));
    if (out_n_mf) *out_n_mf = nmf;
    h->call++;
    return 0;
}

// msre of sigma(mu0 W0^T + vb) against X (dbm.py:625-630), mu from the last mean_field()
static int recon_msre(bm_dbm *h, const float *X_dev, float *out_msre) {
    reconstruct_from_mu(h, h->recon.p, h->recon.ld);
    BM_HIP(hipMemsetAsync(h->scal, 0, sizeof(double), h->stream));
    hipLaunchKernelGGL(sqdiff_kernel, dim3(128), dim3(256), 0, h->stream, X_dev, h->V, (const float *)h->recon.p,
                       h->recon.ld, h->N, h->V, h->scal);
    double s = 0.0;
    BM_HIP(hipMemcpyAsync(&s, h->scal, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
    *out_msre = (float)(s / ((double)h->N * h->V));
    return 0;
}

// session.run([msre, n_mf_updates]) of _run_val_metrics (dbm.py:813): both tensors are built under
// tf.control_dependencies([v_update, v_new_update] + H_updates + H_new_updates + mu_updates) (:521-523),
// so the fetch runs the mean-field on X AND advances the fantasy particles by n_gibbs_steps; no parameter update.
int bm_dbm_metrics(bm_dbm *h, const float *X_dev, int32_t k, int32_t *out_n_mf, float *out_msre) {
    BM_CHECK(k >= 1, "n_gibbs_steps must be >= 1 (got %d)", k);
    int nmf = 0;
    BM_TRY(mean_field_and_particles(h, X_dev, k, &nmf));
    if (out_msre) BM_TRY(recon_msre(h, X_dev, out_msre));
    if (out_n_mf) *out_n_mf = nmf;
    h->call++;
    BM_CHECK(!h->failed, "bm_dbm: a device allocation failed inside a sweep (row store of a Multinomial layer)");
    BM_HIP(hipGetLastError());
    return 0;
}

int bm_dbm_set_comm(bm_dbm *h, bm_comm *c) {
    BM_CHECK(h, "null argument");
    h->comm = c;
    return 0;
}

// Opt-in fast-binary mode (bm_bf3.h): contractions whose input states are {0,1} bitmaps run as exact-product
// bf16 x 3 on the bf16 matrix cores (AIS with all layers sampled).  Results then agree with the default fp32 chain to
// fp32 round-off, not bit for bit.  0 restores the default.  1 = where it pays: AIS (0.82 -> 0.42 s per 1000-beta run of
// 20 000 chains) and the particle sweeps of stacks with >= 8M weights in the bottom layer (3072 x 5000: +4 %); 2 = wherever
// legal (tests, measurements: the particle sweeps of the 784-512-1024 stack are SLOWER in this mode).
int bm_dbm_set_fast_binary(bm_dbm *h, int32_t on) {
    BM_CHECK(h, "null argument");
    BM_CHECK(!(on && h->sigmoid_literal), "fast-binary mode and the literal sigmoid exclude each other");
    h->fast = on >= 2 ? 2 : (on ? 1 : 0);
    return 0;
}

// 1: the AIS log-weights are accumulated as the reference's graph does it - every log p*_beta(x) formed and added /
// subtracted in float32, in the order of dbm.py:708-728 (two extra score-only passes per beta); 0 (default): the
// difference of the two softplus terms per element, summed in double (deterministic, closer to the exactly
// enumerable log Z; the reference's README admits the nats its float32 loop loses at many betas)
int bm_dbm_set_ais_literal(bm_dbm *h, int32_t on) {
    BM_CHECK(h, "null argument");
    h->ais_literal = on ? 1 : 0;
    return 0;
}

// 1: every Bernoulli activation of this engine - mean-field, particle sweeps, AIS transitions, reconstruction - is the literal
// float32 `1 / (1 + exp(-x))` of tf.nn.sigmoid (layers.py:47-48; bm_numerics.h sigmoid_literal) instead of the engine's own
// one-division form.  The values agree to float32 round-off; what changes is the mean-field trip count at an mf_tol near
// that round-off (dbm.py:449-452 at the default 1e-7 is decided in the last bits of the means): with the literal form the
// engine executes the sweeps the reference's graph executes.  Turns the fast-binary mode off for this engine (its epilogue
// is the hardware sigmoid).
int bm_dbm_set_sigmoid_literal(bm_dbm *h, int32_t on) {
    BM_CHECK(h, "null argument");
    h->sigmoid_literal = on ? 1 : 0;
    if (on) h->fast = 0;
    return 0;
}

int bm_dbm_set_xchg(bm_dbm *h, bm_xchg *x) {
    BM_CHECK(h, "null argument");
    h->xchg = x;
    if (x) { h->xchg_used = x; xchg_bind_user(x, &h->xchg_used); }
    return 0;
}

int bm_dbm_set_mf_allreduce(bm_dbm *h, float (*fn)(float, void *), void *ctx) {
    h->mf_reduce = fn; h->mf_ctx = ctx;
    return 0;
}

// data-parallel halves: phase 1 leaves the raw local sums in "grad", the caller
// all-reduces that buffer, phase 2 normalises with the GLOBAL N and M and applies the update.
int bm_dbm_grad_step(bm_dbm *h, const float *X_dev, int32_t k, int32_t *out_n_mf) {
    BM_CHECK(k >= 1, "n_gibbs_steps must be >= 1 (got %d)", k);
    int nmf = 0;
    BM_TRY(mean_field_and_particles(h, X_dev, k, &nmf));
    launch_dbm_colsums(h, X_dev);
    for (int i = 0; i < h->L; ++i) launch_dbm_grad(h, X_dev, i, 0, 1.f, 1.f, 0.f, 0.f);
    if (out_n_mf) *out_n_mf = nmf;
    h->call++;
    BM_CHECK(!h->failed, "bm_dbm: a device allocation failed inside a sweep (row store of a Multinomial layer)");
    BM_HIP(hipGetLastError());
    return 0;
}

int bm_dbm_apply_step(bm_dbm *h, int32_t N_global, int32_t M_global, float lr, float mom) {
    const float N = (float)N_global, M = (float)M_global;
    BM_TRY(check_dw(h, "bm_dbm_apply_step"));
    launch_dbm_biases(h, N, M, lr, mom);
    for (int i = 0; i < h->L; ++i) {
        ApplyWArgs a;
        memset(&a, 0, sizeof(a));
        a.raw = h->grad.p + h->raw_off[i][0]; a.raw2 = h->grad.p + h->raw_off[i][1];
        a.W = h->W[i].p; a.dW = h->dW[i].p; a.Wt = nullptr; a.pen = h->pen[i].p;
        a.I = h->n[i + 1]; a.J = h->n[i]; a.ldw = h->W[i].ld; a.ldwt = h->Wt[i].ld; a.form = 1;
        a.N = N; a.M = M; a.l2 = h->cfg.l2; a.lr = lr; a.mom = mom;
        launch_apply_w(a, nullptr, h->stream);
        launch_dbm_maxnorm(h, i);
    }
    BM_CHECK(!h->failed, "bm_dbm: a device allocation failed inside a sweep (row store of a Multinomial layer)");
    BM_HIP(hipGetLastError());
    return 0;
}

int bm_dbm_stream(bm_dbm *h, void **out_stream) { *out_stream = (void *)h->stream; return 0; }

int bm_dbm_mean_field(bm_dbm *h, const float *X_dev, float *MU_top_dev, int32_t *out_n_mf) {
    int nmf = 0;
    BM_TRY(mean_field(h, X_dev, &nmf));
    if (MU_top_dev) {
        const Mat &t = h->mu[h->L - 1];
        hipLaunchKernelGGL(copy2d_kernel, dim3(256), dim3(256), 0, h->stream, (const float *)t.p, t.ld, MU_top_dev, t.cols,
                           t.rows, t.cols);
    }
    if (out_n_mf) *out_n_mf = nmf;
    h->call++;
    BM_CHECK(!h->failed, "bm_dbm: a device allocation failed inside a sweep (row store of a Multinomial layer)");
    BM_HIP(hipGetLastError());
    return 0;
}

int bm_dbm_reconstruct(bm_dbm *h, const float *X_dev, float *R_dev) {
    BM_CHECK(R_dev, "null output");
    BM_TRY(mean_field(h, X_dev, nullptr));
    reconstruct_from_mu(h, R_dev, h->V);
    h->call++;
    BM_CHECK(!h->failed, "bm_dbm: a device allocation failed inside a sweep (row store of a Multinomial layer)");
    BM_HIP(hipGetLastError());
    return 0;
}

int bm_dbm_sample_v(bm_dbm *h, int32_t k, float *V_dev) {
    BM_CHECK(k >= 0, "n_gibbs_steps must be >= 0");
    particles_update(h, k, true);                             // :643-644
    // `_make_particles_update(sample=False)` whose v assign is the only one fetched (:646-647):
    // k mean sweeps from the sampled state; only v takes the result, H / *_new keep theirs.
    // scratch: reuse mu_alt-sized buffers is not possible (M != N): allocate temporaries
    Mat tv, tv2; std::vector<Mat> tH(h->L), tH2(h->L);
    BM_TRY(tv.alloc(h->M, h->V)); BM_TRY(tv2.alloc(h->M, h->V));
    for (int i = 0; i < h->L; ++i) { BM_TRY(tH[i].alloc(h->M, h->n[i + 1])); BM_TRY(tH2[i].alloc(h->M, h->n[i + 1])); }
    const Mat *Hin = h->H; LayerIn vin{h->v.p, h->v.ld};
    Mat *Hout = tH.data(), *Hout2 = tH2.data(); Mat *vout = &tv, *vout2 = &tv2;
    for (int t = 0; t < k; ++t) {
        gibbs_sweep(h, h->M, vin, Hin, vout, Hout, true, false, k + t, h->prow0);
        vin = LayerIn{vout->p, vout->ld}; Hin = Hout;
        Mat *x = Hout; Hout = Hout2; Hout2 = x;
        Mat *y = vout; vout = vout2; vout2 = y;
    }
    // v <- v_means (the last vout is now vout2 after the swap).  k == 0: no sweep ran, v keeps its value
    // (the reference's op list is empty then, dbm.py:641-648; oracle: orc_dbm_sample_v)
    if (k > 0)
        hipLaunchKernelGGL(copy2d_kernel, dim3(256), dim3(256), 0, h->stream, (const float *)vout2->p, vout2->ld, h->v.p, h->v.ld,
                           h->M, h->V);
    if (V_dev)
        hipLaunchKernelGGL(copy2d_kernel, dim3(256), dim3(256), 0, h->stream, (const float *)h->v.p, h->v.ld, V_dev, h->V,
                           h->M, h->V);
    BM_HIP(hipStreamSynchronize(h->stream));
    tv.release(); tv2.release();
    for (int i = 0; i < h->L; ++i) { tH[i].release(); tH2[i].release(); }
    h->call++;
    return 0;
}

static inline int nslots(int n) { return (n + 15) / 16; }

static int ensure_ais(bm_dbm *h, int rows) {
    if (rows <= h->ais_rows) return 0;
    Mat *ms[] = {&h->ax, &h->ax2, &h->av, &h->ah2};
    for (Mat *m : ms) m->release();
    DevBuf *bs[] = {&h->apart_v, &h->apart_h, &h->apart_x[0], &h->apart_x[1], &h->rowtmp};
    for (DevBuf *b : bs) b->release();
    if (h->alogw) { (void)hipFree(h->alogw); h->alogw = nullptr; }
    const int H2 = h->L >= 2 ? h->n[2] : 1;
    BM_TRY(h->ax.alloc(rows, h->n[1])); BM_TRY(h->ax2.alloc(rows, h->n[1]));
    BM_TRY(h->av.alloc(rows, h->V)); BM_TRY(h->ah2.alloc(rows, H2));
    BM_TRY(h->apart_v.alloc((size_t)nslots(h->V > h->n[1] ? h->V : h->n[1]) * rows));   // AIS: V slots; ELBO: n1 slots
    BM_TRY(h->apart_h.alloc((size_t)nslots(H2 > h->n[1] ? H2 : h->n[1]) * rows));
    BM_TRY(h->apart_x[0].alloc((size_t)nslots(h->n[1]) * rows)); BM_TRY(h->apart_x[1].alloc((size_t)nslots(h->n[1]) * rows));
    BM_TRY(h->rowtmp.alloc(rows));
    BM_HIP(hipMalloc((void **)&h->alogw, (size_t)rows * sizeof(double)));
    h->ax16.release(); h->ax2_16.release(); h->av16.release(); h->ah2_16.release();     // (re)allocated by the fast path
    h->ais_rows = rows;
    return 0;
}

// x0 ~ Bernoulli(1/2): Bernoulli(logits=0).sample(seed) (dbm.py:699-702)
__global__ void ais_init_kernel(float *X, int ld, int rows, int cols, PhiloxKey key, unsigned long long row0) {
    const size_t n = (size_t)rows * cols;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / (size_t)cols, c = e % (size_t)cols;
        X[r * ld + c] = (philox_uniform_at(key, (row0 + r) * (unsigned long long)cols + c) < 0.5f) ? 1.f : 0.f;
    }
}

// rowdot[j] = sum_i X[j][i] * vec[i]  (one wave per row; fixed lane-strided order + butterfly: deterministic)
__global__ __launch_bounds__(256) void rowdot_kernel(const float *X, int ld, int rows, int cols, const float *vec, float *out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += X[(size_t)row * ld + c] * vec[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[row] = s;
}

// One AIS score: logw[j] += sum_slots pv + sum_slots ph + (beta_b - beta_a) * sum_slots pd   (dbm.py:650-660)
// from the slot partials act_kernel left (ActArgs::rowacc / rowdot_out), in double, in a FIXED order: 32 chains per
// workgroup, 8 thread groups per chain; group t adds the slots q = t, t + 8, ... of each of the three partial arrays in
// ascending order (consecutive threads read consecutive chains: full lines), the 8 group sums are added as a fixed
// tree.  (Round 2: one thread per chain walked all 113 slots, 79 workgroups for 20 000 chains: 37 us per beta, 4 % of
// an AIS run.)  Deterministic; sums of <= 113 floats in double are exact to ~1e-16, far below the float the value
// is finally rounded to.
__global__ __launch_bounds__(256) void ais_score_kernel(double *logw, int J, int ld, const float *pv, int nv, const float *ph, int nh,
                                                        const float *pd, int nd, float dbeta) {
    __shared__ double s_s[8][32], s_d[8][32];
    const int c = threadIdx.x & 31, t = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + c;
    double s = 0.0, d = 0.0;
    if (j < J) {
        for (int q = t; q < nv; q += 8) s += (double)pv[(size_t)q * ld + j];
        for (int q = t; q < nh; q += 8) s += (double)ph[(size_t)q * ld + j];
        for (int q = t; q < nd; q += 8) d += (double)pd[(size_t)q * ld + j];
    }
    s_s[t][c] = s; s_d[t][c] = d;
    __syncthreads();
    if (t == 0 && j < J) {
        const double ss = ((s_s[0][c] + s_s[1][c]) + (s_s[2][c] + s_s[3][c])) + ((s_s[4][c] + s_s[5][c]) + (s_s[6][c] + s_s[7][c]));
        const double dd = ((s_d[0][c] + s_d[1][c]) + (s_d[2][c] + s_d[3][c])) + ((s_d[4][c] + s_d[5][c]) + (s_d[6][c] + s_d[7][c]));
        logw[j] += ss + (double)dbeta * dd;
    }
}

// LITERAL accumulation (bm_dbm_set_ais_literal; the reference's arithmetic, dbm.py:650-660 and :708-728): one call
// adds or subtracts ONE log p*_beta(x) to the running log-weight, everything in float32 -
//   lp = (x.hb0 * beta + sum_i softplus(beta (x W0^T + vb)_i)) + sum_k softplus(beta (x W1 + hb1)_k);   lz = lz -/+ lp
// (`T1 *= beta; log_p = T1; log_p += reduce_sum(..); log_p += reduce_sum(..)`; `log_Z += / -= ...`).  The row sums
// are the slot partials added in ascending order in float32.  logw holds the float value (exactly) in its double.
__global__ __launch_bounds__(256) void ais_score_literal_kernel(double *logw, int J, int ld, const float *pv, int nv, const float *ph,
                                                                int nh, const float *pd, int nd, float beta, int sign) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= J) return;
    float sv = 0.f, sh = 0.f, dot = 0.f;
    for (int q = 0; q < nv; ++q) sv = sv + pv[(size_t)q * ld + j];
    for (int q = 0; q < nh; ++q) sh = sh + ph[(size_t)q * ld + j];
    for (int q = 0; q < nd; ++q) dot = dot + pd[(size_t)q * ld + j];
    float lp = dot * beta;
    lp = lp + sv;
    lp = lp + sh;
    const float lz = (float)logw[j];
    logw[j] = (double)(sign > 0 ? lz + lp : lz - lp);
}

// the AIS run itself: leaves the per-chain log-weights (without log Z_0) in h->alogw [n_runs] (device, double)
static int ais_core(bm_dbm *h, int32_t n_betas, int32_t n_runs, int32_t k, uint64_t seed, int64_t chain0) {
    BM_CHECK(h->L == 2, "AIS is implemented for 2-layer DBMs only (dbm.py:925)");
    BM_CHECK(h->cfg.v_unit == BM_UNIT_BERNOULLI, "AIS needs Bernoulli visible units (dbm.py:926-927)");
    BM_CHECK(!h->multinomial(0) && !h->multinomial(1), "AIS needs Bernoulli hidden layers (dbm.py:926-927)");
    BM_CHECK(n_betas >= 2 && n_runs >= 1 && k >= 1, "bad AIS arguments");
    BM_TRY(ensure_ais(h, n_runs));
    const int R = n_runs, V = h->V, H1 = h->n[1], H2 = h->n[2];
    const float db = 1.0f / (float)n_betas;                               // delta_beta (dbm.py:929)
    // x.hb0 of the current / next state as slot partials (pitch ldp); x0's comes from rowdot_kernel as ONE slot
    const int ldp = h->ais_rows;
    float *rdot_cur = h->apart_x[0].p, *rdot_next = h->apart_x[1].p;
    int nd_cur = 1;
    BM_HIP(hipMemsetAsync(h->alogw, 0, (size_t)R * sizeof(double), h->stream));
    Mat *x = &h->ax, *xn = &h->ax2;
    hipLaunchKernelGGL(ais_init_kernel, dim3(512), dim3(256), 0, h->stream, x->p, x->ld, R, H1,
                       dkey(h, SITE_AIS_X0, 0, seed, 0), (unsigned long long)chain0);
    // fast-binary mode: every state of the run is a {0,1} bitmap when all three layers are sampled (the default)
    struct FastScope { bm_dbm *h; ~FastScope() { h->fast_now = false; } } fast_scope{h};
    if (h->fast && h->cfg.sample_v_states && h->cfg.sample_h_states[0] && h->cfg.sample_h_states[1]) {
        BM_TRY(fast_build_planes(h));
        if (h->ax16.rows != h->ais_rows) {
            BM_TRY(h->ax16.alloc(1, h->ais_rows, H1)); BM_TRY(h->ax2_16.alloc(1, h->ais_rows, H1));
            BM_TRY(h->av16.alloc(1, h->ais_rows, V)); BM_TRY(h->ah2_16.alloc(1, h->ais_rows, H2));
        }
        hipLaunchKernelGGL(shadow16_kernel, dim3(512), dim3(256), 0, h->stream, (const float *)x->p, x->ld, R, H1, h->ax16.p, h->ax16.ld);
        h->fast_now = true; h->fast_ais = true;
    }
    hipLaunchKernelGGL(rowdot_kernel, dim3((R + 3) / 4), dim3(256), 0, h->stream, (const float *)x->p, x->ld, R, H1,
                       (const float *)h->hb[0].p, rdot_cur);

    // visit(x; beta_a, beta_b, beta_c): logw += log p*_{beta_b}(x) - log p*_{beta_a}(x) (score != 0),
    // then k transitions T_{beta_c} (dbm.py:662-694).  `step` feeds the RNG call counter.
    const bool literal = h->ais_literal != 0;
    // literal mode: -log p*_{ba}(x) as its own pair of score-only launches (the default path shares the pre-activations
    // of the transition and accumulates the DIFFERENCE of the two softplus terms per element, in double)
    auto score_only = [&](float bscore, int sign) -> int {
        ActArgs e;
        memset(&e, 0, sizeof(e));
        e.rowacc = h->apart_v.p; e.ld_part = ldp; e.beta_b = bscore; e.rowacc_single = 1;
        layer_update(h, -1, R, LayerIn{nullptr, 0}, LayerIn{x->p, x->ld}, bscore, bscore, 0, nullptr, nullptr, h->av.ld,
                     dkey(h, SITE_DBM_V, 0, seed, 0), chain0, nullptr, nullptr, &e);
        memset(&e, 0, sizeof(e));
        e.rowacc = h->apart_h.p; e.ld_part = ldp; e.beta_b = bscore; e.rowacc_single = 1;
        layer_update(h, 1, R, LayerIn{x->p, x->ld}, LayerIn{nullptr, 0}, bscore, bscore, 0, nullptr, nullptr, h->ah2.ld,
                     dkey(h, SITE_DBM_H + 1, 0, seed, 0), chain0, nullptr, nullptr, &e);
        hipLaunchKernelGGL(ais_score_literal_kernel, dim3((R + 255) / 256), dim3(256), 0, h->stream, h->alogw, R, ldp,
                           (const float *)h->apart_v.p, nslots(V), (const float *)h->apart_h.p, nslots(H2),
                           (const float *)rdot_cur, nd_cur, bscore, sign);
        return 0;
    };
    auto visit = [&](bool score, float ba, float bb, bool transit, float bc, uint32_t step) -> int {
        if (literal && score) {                 // log_Z -= log p*_{ba}(x); log_Z += log p*_{bb}(x)   (:708, :714, :718, :728)
            BM_TRY(score_only(ba, -1));
            BM_TRY(score_only(bb, +1));
            score = false;
        }
        for (int t = 0; t < (transit ? k : 1); ++t) {
            const bool sc = score && t == 0;
            ActArgs e;
            // v~ <- P(v | h = x): sigma(beta*x W0^T + beta*vb)  — and the visible softplus term of log p*
            memset(&e, 0, sizeof(e));
            if (sc) { e.rowacc = h->apart_v.p; e.ld_part = ldp; e.beta_a = ba; e.beta_b = bb; }
            const int smp_v = transit && h->cfg.sample_v_states;
            layer_update(h, -1, R, LayerIn{nullptr, 0}, LayerIn{x->p, x->ld}, bc, bc, smp_v,
                         (transit && !smp_v) ? h->av.p : nullptr, (transit && smp_v) ? h->av.p : nullptr, h->av.ld,
                         dkey(h, SITE_DBM_V, t, seed, step), chain0, nullptr, nullptr, &e);
            // h2~ <- P(h2 | h = x): sigma(beta*x W1 + beta*hb1)  — and the top softplus term
            memset(&e, 0, sizeof(e));
            if (sc) { e.rowacc = h->apart_h.p; e.ld_part = ldp; e.beta_a = ba; e.beta_b = bb; }
            const int smp_2 = transit && h->cfg.sample_h_states[1];
            layer_update(h, 1, R, LayerIn{x->p, x->ld}, LayerIn{nullptr, 0}, bc, bc, smp_2,
                         (transit && !smp_2) ? h->ah2.p : nullptr, (transit && smp_2) ? h->ah2.p : nullptr, h->ah2.ld,
                         dkey(h, SITE_DBM_H + 1, t, seed, step), chain0, nullptr, nullptr, &e);
            if (sc)     // both softplus terms + (bb - ba) * x.hb0, slots in fixed order, into the double log-weights
                hipLaunchKernelGGL(ais_score_kernel, dim3((R + 31) / 32), dim3(256), 0, h->stream, h->alogw, R, ldp,
                                   (const float *)h->apart_v.p, nslots(V), (const float *)h->apart_h.p, nslots(H2),
                                   (const float *)rdot_cur, nd_cur, bb - ba);
            if (!transit) break;
            // x^ <- P(h | v~, h2~): sigma(beta*(v W0 + h2 W1^T) + beta*hb0); also x^.hb0 for the next score
            memset(&e, 0, sizeof(e));
            e.rowdot_out = rdot_next; e.ld_part = ldp; e.dot_vec = h->hb[0].p;
            const int smp_x = h->cfg.sample_h_states[0];
            layer_update(h, 0, R, LayerIn{h->av.p, h->av.ld}, LayerIn{h->ah2.p, h->ah2.ld}, bc, bc, smp_x,
                         nullptr, xn->p, xn->ld, dkey(h, SITE_DBM_H + 0, t, seed, step), chain0, nullptr, nullptr, &e);
            Mat *tm = x; x = xn; xn = tm;
            float *tr = rdot_cur; rdot_cur = rdot_next; rdot_next = tr;
            nd_cur = nslots(H1);
        }
        return 0;
    };
    // x_1 ~ T_{db}(x_0)                                                     (:704-705)
    BM_TRY(visit(false, 0.f, 0.f, true, db, 0));
    // -log p_0(x_1), then the loop over beta = db, 2db, ... (fp32 accumulation, :710-726)
    float beta = db, prev = 0.f;
    uint32_t step = 1;
    while (beta < 1.0f - db + 1e-5f) {
        BM_TRY(visit(true, prev, beta, true, beta + db, step++));            // +log p_beta(x) -log p_prev(x); x' ~ T_{beta+db}
        prev = beta;
        beta = beta + db;
    }
    BM_TRY(visit(true, prev, 1.0f, false, 0.f, step++));                     // +log p_1(x_M) - log p_prev(x_M)  (:728)
    BM_CHECK(!h->failed, "bm_dbm: a device allocation failed inside a sweep (row store of a Multinomial layer)");
    BM_HIP(hipGetLastError());
    return 0;
}

static double ais_log_Z0(const bm_dbm *h) {                                  // (:731-734)
    return (double)(h->V + h->n[1] + h->n[2]) * (double)logf(2.0f);
}

int bm_dbm_ais(bm_dbm *h, int32_t n_betas, int32_t n_runs, int32_t k, uint64_t seed, int64_t chain0,
               float *values_host) {
    BM_CHECK(values_host, "null output");
    BM_TRY(ais_core(h, n_betas, n_runs, k, seed, chain0));
    const int R = n_runs;
    std::vector<double> w(R);
    BM_HIP(hipMemcpyAsync(w.data(), h->alogw, (size_t)R * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
    const double logZ0 = ais_log_Z0(h);
    if (h->ais_literal) {                        // log_Z += log_Z0 in float32 (:731-734)
        const float z0 = (float)(h->V + h->n[1] + h->n[2]) * logf(2.0f);
        for (int r = 0; r < R; ++r) values_host[r] = (float)w[r] + z0;
    } else {
        for (int r = 0; r < R; ++r) values_host[r] = (float)(w[r] + logZ0);
    }
    return 0;
}

// values[r] = (float)(logw[r] + log Z_0) for r < n, 0 in the padding up to npad
__global__ void ais_finish_kernel(const double *logw, float *out, int n, int npad, double logZ0, int literal) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < npad) out[r] = (r < n) ? (literal ? (float)logw[r] + (float)logZ0 : (float)(logw[r] + logZ0)) : 0.f;
}

// Chain-sharded AIS: this rank runs chains [start, stop) of n_runs_total (contiguous slices, the
// remainder spread over the first ranks), no communication during the sweep, then ONE all-gather of the
// per-chain values over the library's communicator; every rank returns all n_runs_total values.
int bm_dbm_ais_sharded(bm_dbm *h, bm_comm *c, int32_t n_betas, int32_t n_runs_total, int32_t k, uint64_t seed,
                       float *values_host) {
    BM_CHECK(h && c && values_host, "null argument");
    int32_t rank = 0, world = 1;
    BM_TRY(bm_comm_rank(c, &rank, &world));
    BM_CHECK(n_runs_total >= 1, "bad AIS arguments");
    auto shard = [&](int r, int &a, int &b) {
        const int q = n_runs_total / world, rem = n_runs_total % world;
        a = r * q + (r < rem ? r : rem);
        b = a + q + (r < rem ? 1 : 0);
    };
    int a = 0, b = 0;
    shard(rank, a, b);
    const int n = b - a, npad = (n_runs_total + world - 1) / world;
    // send / receive buffers live in the handle (grown on demand, never on the steady path)
    if (h->ais_send.n < (size_t)npad) { h->ais_send.release(); BM_TRY(h->ais_send.alloc((size_t)npad)); }
    if (h->ais_recv.n < (size_t)npad * world) { h->ais_recv.release(); BM_TRY(h->ais_recv.alloc((size_t)npad * world)); }
    float *send = h->ais_send.p, *recv = h->ais_recv.p;
    // A rank whose sweep fails must still enter the collective - the other ranks would block in it forever - so the
    // failure is made collective: the failing rank contributes NaNs and every rank reports the error.
    std::string first_err;
    int rc = 0;
    if (n > 0) rc = ais_core(h, n_betas, n, k, seed, a);
    if (rc) {
        first_err = bm_last_error();
        (void)hipGetLastError();
        std::vector<float> nan((size_t)npad, __builtin_nanf(""));
        (void)hipMemcpyAsync(send, nan.data(), nan.size() * sizeof(float), hipMemcpyHostToDevice, h->stream);
        (void)hipStreamSynchronize(h->stream);
    } else {
        const double z0 = h->ais_literal ? (double)((float)(h->V + h->n[1] + h->n[2]) * logf(2.0f)) : ais_log_Z0(h);
        hipLaunchKernelGGL(ais_finish_kernel, dim3((npad + 255) / 256), dim3(256), 0, h->stream, (const double *)h->alogw, send,
                           n, npad, z0, h->ais_literal);
    }
    const int rc_c = bm_comm_allgather(c, send, recv, (size_t)npad, (void *)h->stream);
    if (rc_c && first_err.empty()) first_err = bm_last_error();
    std::vector<float> all((size_t)npad * world);
    int rc_m = 0;
    if (!rc_c && hipMemcpyAsync(all.data(), recv, all.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc_m = 1;
    if (hipStreamSynchronize(h->stream) != hipSuccess) rc_m = 1;
    if (rc || rc_c) { bm::set_error("bm_dbm_ais_sharded (rank %d): %s", rank, first_err.c_str()); return rc ? rc : rc_c; }
    if (rc_m) { bm::set_error("bm_dbm_ais_sharded: device copy / synchronisation failed"); return 1; }
    bool peer_failed = false;
    for (int r = 0; r < world; ++r) {
        int ra, rb;
        shard(r, ra, rb);
        memcpy(values_host + ra, all.data() + (size_t)r * npad, (size_t)(rb - ra) * sizeof(float));
        for (int e = ra; e < rb; ++e) if (values_host[e] != values_host[e]) { peer_failed = true; break; }
    }
    BM_CHECK(!peer_failed, "bm_dbm_ais_sharded: another rank's AIS sweep failed (its slice arrived as NaN)");
    return 0;
}

// per-row bias terms and entropies of the ELBO (dbm.py:746-756), one wave per row; p0 / p1: slot partials of
// sum((X W0) * mu0) and sum((mu0 W1) * mu1) from the two propagations (pitch ldp), added in slot order
__global__ __launch_bounds__(256) void elbo_row_kernel(const float *X, int ldx, int V, const float *vb,
                                                       const float *mu0, int ld0, int H1, const float *hb0,
                                                       const float *mu1, int ld1, int H2, const float *hb1,
                                                       int rows, const float *p0, int n0, const float *p1, int n1, int ldp,
                                                       float *out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < V; c += 64) s += X[(size_t)row * ldx + c] * vb[c];
    for (int c = lane; c < H1; c += 64) {
        const float m = mu0[(size_t)row * ld0 + c];
        s += m * hb0[c];
        const float q = fminf(fmaxf(m, 1e-7f), 1.f - 1e-7f);
        s += -q * logf(q) - (1.f - q) * logf(1.f - q);
    }
    for (int c = lane; c < H2; c += 64) {
        const float m = mu1[(size_t)row * ld1 + c];
        s += m * hb1[c];
        const float q = fminf(fmaxf(m, 1e-7f), 1.f - 1e-7f);
        s += -q * logf(q) - (1.f - q) * logf(1.f - q);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
        double e = 0.0;
        for (int q = 0; q < n0; ++q) e += (double)p0[(size_t)q * ldp + row];
        for (int q = 0; q < n1; ++q) e += (double)p1[(size_t)q * ldp + row];
        out[row] = (float)(e + (double)s);
    }
}

int bm_dbm_log_proba(bm_dbm *h, const float *X_dev, float *out_host) {
    BM_CHECK(h->L == 2, "log_proba is implemented for 2-layer DBMs only (dbm.py:741-756)");
    BM_CHECK(!h->multinomial(0) && !h->multinomial(1), "log_proba needs Bernoulli hidden layers (dbm.py:947-948)");
    BM_CHECK(out_host, "null output");
    BM_TRY(mean_field(h, X_dev, nullptr));
    BM_TRY(ensure_ais(h, h->N));
    // sum((X W0) * mu0) and sum((mu0 W1) * mu1) as dot-epilogues of the two propagations (slot partials)
    const int ldp = h->ais_rows;
    ActArgs e;
    memset(&e, 0, sizeof(e));
    e.rowacc = h->apart_v.p; e.ld_part = ldp; e.dot_mat = h->mu[0].p; e.ld_dot = h->mu[0].ld;
    layer_update(h, 0, h->N, LayerIn{X_dev, h->V}, LayerIn{nullptr, 0}, 1.f, 1.f, 0, nullptr, nullptr, h->mu[0].ld,
                 dkey(h, 0, 0, h->seed, h->call), 0, nullptr, nullptr, &e);
    memset(&e, 0, sizeof(e));
    e.rowacc = h->apart_h.p; e.ld_part = ldp; e.dot_mat = h->mu[1].p; e.ld_dot = h->mu[1].ld;
    layer_update(h, 1, h->N, LayerIn{h->mu[0].p, h->mu[0].ld}, LayerIn{nullptr, 0}, 1.f, 1.f, 0, nullptr, nullptr,
                 h->mu[1].ld, dkey(h, 0, 0, h->seed, h->call), 0, nullptr, nullptr, &e);
    hipLaunchKernelGGL(elbo_row_kernel, dim3((h->N + 3) / 4), dim3(256), 0, h->stream, X_dev, h->V, h->V,
                       (const float *)h->vb.p, (const float *)h->mu[0].p, h->mu[0].ld, h->n[1], (const float *)h->hb[0].p,
                       (const float *)h->mu[1].p, h->mu[1].ld, h->n[2], (const float *)h->hb[1].p, h->N,
                       (const float *)h->apart_v.p, nslots(h->n[1]), (const float *)h->apart_h.p, nslots(h->n[2]), ldp, h->rowtmp.p);
    BM_HIP(hipMemcpyAsync(out_host, h->rowtmp.p, (size_t)h->N * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
    h->call++;
    return 0;
}

int bm_dbm_timer_start(bm_dbm *h) { BM_HIP(hipEventRecord(h->ev0, h->stream)); return 0; }
// the two halves of timer_stop: the mark is enqueued inside a timed region, the read (a host wait) after it
int bm_dbm_timer_mark(bm_dbm *h) { BM_HIP(hipEventRecord(h->ev1, h->stream)); return 0; }
int bm_dbm_timer_elapsed(bm_dbm *h, float *out_ms) {
    BM_HIP(hipEventSynchronize(h->ev1));
    BM_HIP(hipEventElapsedTime(out_ms, h->ev0, h->ev1));
    return 0;
}
int bm_dbm_timer_stop(bm_dbm *h, float *out_ms) {
    BM_HIP(hipEventRecord(h->ev1, h->stream));
    BM_HIP(hipEventSynchronize(h->ev1));
    BM_HIP(hipEventElapsedTime(out_ms, h->ev0, h->ev1));
    return 0;
}

}  // extern "C"
