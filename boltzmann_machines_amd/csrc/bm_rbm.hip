// bm_rbm.hip — C-ABI entry points for the RBM path (include/bm355.h) and the
// host-side sequencing of the fused kernels for one CD-k update.
//
// Reference graph restated: boltzmann_machines/rbm/base_rbm.py:415-525
// (train op + metrics), :329-413 (propagations, Gibbs chain), rbm/rbm.py:17-22,
// :101-116 (free energies, Gaussian input scaling), layers.py:39-51,73-89.
#include "../../include/bm355.h"
#include "bm_common.h"
#include "bm_kernels.h"
#include "bm_chain.h"

#include <math.h>
#include <atomic>

namespace bm {

static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// RNG site ids (counter word 2 = site + 16 * gibbs_step); DESIGN.md "RNG"
enum : uint32_t { SITE_DROPOUT = 1, SITE_H0 = 2, SITE_V = 3, SITE_H = 4, SITE_PLL = 5, SITE_FE = 6 };

}  // namespace bm

using namespace bm;

struct bm_xchg;
// (bm_xchg.hip, later in this translation unit)
static int xchg_check_status(bm_xchg *x);                    // error when a wait of the exchange ever expired
static void xchg_bind_user(bm_xchg *x, bm_xchg **slot);      // the engine field that points at x (cleared by bm_xchg_destroy)
static void xchg_dw_replaced(bm_xchg *x);                    // every replica's dW was overwritten whole (set_param)

struct bm_rbm {
    bm_rbm_config cfg;
    bm_xchg *xchg_used = nullptr;     // the direct exchange this engine's gradients last went through (bm_rbm_sync checks it)
    // set by bm_rbm_exchange_apply_direct on more than one rank: every rank then holds only ITS slice of the momentum
    // buffer dW.  Cleared by bm_rbm_exchange_gather_dw (collective) and by a set_param of dW.  While it is set every
    // reader of dW - get_param, stage (checkpoints), apply_step and the single-GPU fused update - fails (check_dw)
    bool dw_sharded = false;
    int V, H, maxB;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // variables (padded pitch, see pad_ld).  The prop-down reads W itself as an x-major operand (ActArgs::p_xm).
    Mat W, dW;                         // [V][H], [V][H]
    // the transpose [H][V], written by the fused update next to W: the prop-up then reads its weights x-major as well
    // (one ds_read_b128 per 16 k and lane instead of four ds_read_b32): 12.95 -> 12.5 us per prop-up, +0.45 us in the
    // update's epilogue (3.2 MB more stores): 64.1 -> 63.7 us per CD-1 update, same box, alternating runs
    // (tools/upxm_ab.sh; BM355_DEBUG=up_xm=0 switches it off).  Valid only while every write of W went through that kernel
    // (wt_valid: cleared by set_param, apply_step and the exchange); otherwise the prop-up reads W k-major as before.
    Mat Wt;
    bool use_wt = false, wt_valid = false;
    DevBuf vb, hb, dvb, dhb, q, sigma;
    // chain workspaces
    Mat h0m, h0s, hm, hs, hneg;        // [maxB][H]; hneg = -hm (negative-phase operand of the outer products)
    Mat vm, vs, Xs, Xd;                // [maxB][V]
    DevBuf grad;      // [V*ldH | V | H | H] raw sums (the data-parallel all-reduce buffer): the ACTIVE one of two
    // delayed-gradient data parallelism (bm_rbm_set_grad_slot / _allreduce_grads_async / _wait_grads): the buffer of
    // the other slot, the stream the reductions run on and their events; allocated at the first use of slot 1
    DevBuf grad_alt;
    int grad_slot = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_reduced[2] = {nullptr, nullptr};
    DevBuf pen;       // [H]
    DevBuf rowacc;    // [3*maxB]
    DevBuf hhat;      // [3*H] MultinomialRBM free-energy h_hat vectors (rbm.py:58)
    int *flip = nullptr;
    double *scal = nullptr;   // [6] device accumulators: msre, l2, F(x), F(x~), F'(x) (multinomial), spare
    bool multinomial() const { return cfg.h_unit == BM_UNIT_MULTINOMIAL; }
    uint64_t seed = 0;
    uint32_t call = 0;
    int64_t row0 = 0;
    // input of the last run_chain() (after /sigma and dropout) and its pitch
    const float *Xin = nullptr;
    int Xin_ld = 0;
    bool hm_is_neg = false;    // the last run_chain() wrote -h_k (hneg) instead of h_k (hm)
    bool fe_in_chain = false;  // the last run_chain() left the free-energy slot partials of its input in fe_part (metric fetch)
    DevBuf fe_part;            // [2][ceil(H/16)][maxB]: slot partials of sum softplus for x and for its PLL partner
    // fast-binary mode (bm_bf3.h, bm_rbm_set_fast_binary): bf16 planes of W ([V][H]: the prop-down operand) and of
    // W^T ([H][V]: the prop-up operand), bf16 shadows of the state workspaces hs / vs; `fast_now` while a sweep with
    // {0,1} states on both sides runs (bm_rbm_gibbs)
    int fast = 0;
    bool fast_now = false;
    Mat16 W3, W3t, hs16, vs16;
    int *nonbinary = nullptr;  // device flag: a state handed to the fast path was not a {0,1} bitmap
    // bm_rbm_stage / bm_rbm_get_staged: device-side copies of every variable taken in stream order (a checkpoint
    // snapshot that does not stop the stream), read back on their own stream by whoever writes the checkpoint
    struct Stage { Mat W, dW; DevBuf vb, hb, dvb, dhb, q, sigma; hipEvent_t ev = nullptr; bool ready = false;
                   std::atomic<int> readers{0}; } stage[2];
    hipStream_t stage_stream = nullptr;
    int last_stage = -1;
    float *stage_host = nullptr;     // pinned bounce buffer of bm_rbm_get_staged ([V][H])
    int device = 0;
    // bm_rbm_train_step_metrics_async: pinned ring of the six device sums of every pending metrics fetch
    static constexpr int MRING = 4096;
    double *mring = nullptr, *mring_dev = nullptr;    // pinned host ring and its device-side address
    std::vector<int> mring_B;
    int mring_n = 0;
    hipEvent_t ev_mlast = nullptr;   // behind the last pending fetch: bm_rbm_collect_metrics waits for IT, not for the stream -
                                     // updates queued after the fetch (the next epoch's first run) keep the device busy meanwhile
    // optional per-kernel-class event timing
    bool prof = false;
    struct Rec { int cls; hipEvent_t a, b; };
    std::vector<Rec> recs;
    size_t grad_tail() const { return (size_t)V * W.ld; }
    // a run of dependent propagation passes recorded by launch_up / launch_down and issued as ONE launch (bm_chain.h)
    ChainState chain;
};

enum { KC_UP = 0, KC_DOWN = 1, KC_GRAD = 2, KC_COLSUM = 3, KC_BIAS = 4, KC_OTHER = 5 };

// the momentum buffer is complete on this rank (see bm_rbm::dw_sharded)
static int check_dw(const bm_rbm *h, const char *what) {
    BM_CHECK(!h->dw_sharded, "%s: after bm_rbm_exchange_apply_direct this rank holds only its slice of the momentum buffer dW; "
             "every rank must call bm_rbm_exchange_gather_dw first (DirectExchange.gather_dw())", what);
    return 0;
}

struct ProfScope {
    bm_rbm *h; hipEvent_t b = nullptr;
    ProfScope(bm_rbm *h_, int cls) : h(h_) {
        if (!h->prof) return;
        hipEvent_t a;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a, h->stream);
        h->recs.push_back({cls, a, b});
    }
    ~ProfScope() { if (b) (void)hipEventRecord(b, h->stream); }
};

static PhiloxKey make_key(const bm_rbm *h, uint32_t site, int t) {
    PhiloxKey k;
    k.k0 = (uint32_t)h->seed;
    k.k1 = (uint32_t)(h->seed >> 32);
    k.site = site + 16u * (uint32_t)t;
    k.call = h->call;
    return k;
}

// a propagation pass: launched now, or recorded while a chained run is open (chain_begin .. chain_end)
static void act_pass(bm_rbm *h, const ActArgs &a, const Operand *alt_p = nullptr) {
    if (h->chain.on) {
        h->chain.rec.push_back(a);
        h->chain.alt_p.resize(h->chain.rec.size(), Operand{nullptr, 0, 0, 0});
        if (alt_p) h->chain.alt_p.back() = *alt_p;
    } else if (alt_p) {
        ActArgs b = a;
        b.P1 = *alt_p; b.p_xm = 1;
        launch_act(b, h->stream);
    } else launch_act(a, h->stream);
}
static void chain_begin(bm_rbm *h) {
    // per-class event timing, Multinomial hidden units (a softmax launch between the passes) and the fast-binary sweep
    // keep their per-pass launches
    h->chain.on = !h->prof && !h->multinomial() && !h->fast_now && chain_mode(h->chain) > 0;
}
static int chain_end(bm_rbm *h) {
    BM_CHECK(chain_flush(h->chain, h->stream, h->maxB) == 0, "chained launch: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}

// W^T for the x-major prop-up when W was last written by something else than the fused update (set_param, a sampling-
// only handle): one transpose, valid until the next such write.  NOT called on the split (data-parallel) step, whose
// apply / exchange rewrites W every step.
static void ensure_wt(bm_rbm *h) {
    if (!h->use_wt || h->wt_valid) return;
    const int nt = ((h->V + 31) / 32) * ((h->H + 31) / 32);
    hipLaunchKernelGGL(transpose_kernel, dim3(nt), dim3(256), 0, h->stream, (const float *)h->W.p, h->W.ld, h->Wt.p, h->Wt.ld, h->V, h->H);
    h->wt_valid = true;
}

// E[h|v] (+ sample): base_rbm.py:339-351.  v [B][V] pitch ldv
static void launch_up(bm_rbm *h, const float *v, int ldv, int B, float *means, float *states, int ldo,
                      int sample, uint32_t site, int t, float *negmeans = nullptr, bool fe = false) {
    ProfScope _ps(h, KC_UP);
    ActArgs a;
    memset(&a, 0, sizeof(a));
    a.P1 = make_operand(h->W.p, h->W.ld, h->H);   // W[k=v][i=h], KM
    a.Q1 = make_operand(v, ldv, B);               // v[j=b][k=v], XM
    a.K1 = h->V;
    a.I = h->H; a.J = B;
    a.bias = h->hb.p; a.sigma = nullptr;
    a.mult = 1.0f + (h->cfg.dbm_first ? 1.0f : 0.0f);
    a.bmult = a.mult;
    a.kind = BM_UNIT_BERNOULLI;
    a.sample = states ? sample : 0;               // no consumer of the states: no draw
    a.means = means; a.states = states; a.negmeans = negmeans; a.ldo = ldo;
    a.key = make_key(h, site, t);
    a.row0 = h->row0;
    if (fe) {                                    // metric fetch: the free-energy row sums from this pass's pre-activations
        const int nslot = (h->H + 15) / 16, rm = (nslot + 3) & ~3;
        a.rowacc = h->fe_part.p; a.rowacc_single = 1; a.beta_b = 1.f; a.ld_part = h->maxB; a.fe_rm = rm;
        a.fe_rowacc2 = h->fe_part.p + (size_t)rm * h->maxB;
        // the flip columns straight from their Philox stream and the zeroing of the six accumulators ride on this pass: the
        // fused fetch has no prep launch
        a.fe_flip = FE_FLIP_FROM_KEY; a.fe_key = make_key(h, SITE_PLL, 0); a.fe_zero = h->scal;
        a.fe_x = v; a.fe_ldx = ldv; a.fe_w = h->W.p; a.fe_ldw = h->W.ld;
    }
    if (h->fast_now && v == h->vs.p) {           // fast-binary: W^T planes x the bf16 shadow of the visible bitmap
        a.b3.P1 = Bf3Operand{h->W3t.p, h->W3t.plane_stride(), h->W3t.ld, h->H};
        a.b3.Q1 = Bf3Operand{h->vs16.p, 0, h->vs16.ld, B};
        a.b3.K1 = h->vs16.ld;
        if (states == h->hs.p) { a.states16 = h->hs16.p; a.ld16 = h->hs16.ld; }
    }
    if (h->multinomial()) {
        // MultinomialLayer (layers.py:54-70): logits from the GEMM, then one wave per row for the
        // softmax (activation) and the multinomial counts (sample)
        if (!means) { means = h->hm.p; ldo = h->hm.ld; }     // pure sampling sweep: hm is the scratch row store
        a.kind = 3; a.sample = 0; a.means = means; a.ldo = ldo; a.states = nullptr; a.negmeans = nullptr;
        launch_act(a, h->stream);
        SmArgs m;
        memset(&m, 0, sizeof(m));
        m.L = means; m.ld = ldo; m.I = h->H; m.J = B; m.M = h->cfg.n_samples; m.sample = sample;
        m.states = states; m.negmeans = negmeans; m.key = a.key; m.row0 = h->row0;
        hipLaunchKernelGGL(softmax_multinomial_kernel, dim3(B), dim3(64), 2 * (size_t)h->H * sizeof(float), h->stream, m);
        return;
    }
    if (h->use_wt && h->wt_valid && !(h->fast_now && v == h->vs.p)) {
        const Operand wt = make_operand(h->Wt.p, h->Wt.ld, h->H);      // W^T[i=h][k=v], x-major: the per-pass launch's P
        act_pass(h, a, &wt);
    } else act_pass(h, a);
}

// E[v|h] (+ sample): base_rbm.py:353-365.  hs [B][H] pitch ldh
static void launch_down(bm_rbm *h, const float *hs, int ldh, int B, float *means, float *states, int ldo,
                        int sample, uint32_t site, int t) {
    ProfScope _ps(h, KC_DOWN);
    ActArgs a;
    memset(&a, 0, sizeof(a));
    a.P1 = make_operand(h->W.p, h->W.ld, h->V);    // W[i=v][k=h], x-major P
    a.p_xm = 1;
    a.Q1 = make_operand(hs, ldh, B);               // h[j=b][k=h], XM
    a.K1 = h->H;
    a.I = h->V; a.J = B;
    a.bias = h->vb.p; a.sigma = h->sigma.p;
    a.mult = 1.0f + (h->cfg.dbm_last ? 1.0f : 0.0f);
    a.bmult = a.mult;
    a.kind = h->cfg.v_unit;
    a.sample = sample;
    a.means = means; a.states = states; a.ldo = ldo;
    a.key = make_key(h, site, t);
    a.row0 = h->row0;
    if (h->fast_now && hs == h->hs.p) {          // fast-binary: W planes x the bf16 shadow of the hidden bitmap
        a.b3.P1 = Bf3Operand{h->W3.p, h->W3.plane_stride(), h->W3.ld, h->V};
        a.b3.Q1 = Bf3Operand{h->hs16.p, 0, h->hs16.ld, B};
        a.b3.K1 = h->hs16.ld;
        if (states == h->vs.p) { a.states16 = h->vs16.p; a.ld16 = h->vs16.ld; }
    }
    act_pass(h, a);
}

// input preprocessing + h0 + k Gibbs steps (base_rbm.py:417-426). Leaves
// h0m/h0s, vm/vs (last step), hm/hs (last step) and Xin in the handle.
// If hm_out != null the last step's h_means are written there (dense, pitch H).
// need_vm: the last step's visible MEANS are wanted (msre metric); a plain update only consumes the
// visible states, and nothing consumes the hidden STATES of the last step: those stores (and their
// share of the kernel-boundary L2 writeback) are skipped.
static bool metrics_fused_ok(const bm_rbm *h);
static void metrics_prep(bm_rbm *h, int B);
static int run_chain(bm_rbm *h, const float *X_dev, int B, int k, float *hm_out, bool need_vm = true, bool for_update = false,
                     bool split_step = false, bool fetch = false) {
    BM_CHECK(B >= 1 && B <= h->maxB, "batch %d outside [1, max_batch=%d]", B, h->maxB);
    BM_CHECK(k >= 1, "n_gibbs_steps must be >= 1 (got %d)", k);
    if (!split_step) ensure_wt(h);
    const float *Xin = X_dev;
    int ldx = h->V;
    if (h->cfg.v_unit == BM_UNIT_GAUSSIAN) {   // rbm.py:107
        hipLaunchKernelGGL(div_cols_kernel, dim3(256), dim3(256), 0, h->stream, Xin, ldx, h->sigma.p, h->Xs.p,
                           h->Xs.ld, B, h->V);
        Xin = h->Xs.p; ldx = h->Xs.ld;
    }
    if (h->cfg.dropout >= 0.f) {               // base_rbm.py:417-418
        hipLaunchKernelGGL(dropout_kernel, dim3(256), dim3(256), 0, h->stream, Xin, ldx, h->Xd.p, h->Xd.ld, B, h->V,
                           h->cfg.dropout, make_key(h, SITE_DROPOUT, 0),
                           (unsigned long long)h->row0 * (unsigned long long)h->V);
        Xin = h->Xd.p; ldx = h->Xd.ld;
    }
    h->Xin = Xin; h->Xin_ld = ldx;
    // a metrics iteration: the flip columns and the zeroed accumulators first, then the h0 pass adds the free-energy row
    // sums of x and of its PLL partner from its own pre-activations (ActArgs::fe_flip) - no GEMM of their own
    const bool fe = fetch && metrics_fused_ok(h);
    if (fe && !h->fe_part.p) BM_TRY(h->fe_part.alloc((size_t)2 * ((((h->H + 15) / 16) + 3) & ~3) * h->maxB));
    h->fe_in_chain = fe;
    if (fetch && !fe) metrics_prep(h, B);
    chain_begin(h);               // h0 and the k Gibbs steps: one launch where the shape allows it (bm_chain.h)
    launch_up(h, Xin, ldx, B, h->h0m.p, h->h0s.p, h->h0m.ld, 1, SITE_H0, 0, nullptr, fe);      // :421-422
    const float *hstate = h->cfg.sample_h_states ? h->h0s.p : h->h0m.p;           // :423
    for (int t = 0; t < k; ++t) {                                                 // :367-378
        const bool last = t == k - 1;
        launch_down(h, hstate, h->hs.ld, B, (need_vm && last) ? h->vm.p : nullptr, h->vs.p, h->vm.ld,
                    h->cfg.sample_v_states, SITE_V, t);
        const bool last_out = hm_out && last;
        // plain update, last step: only -h_k is consumed (outer products and column sums)
        const bool neg_only = last && !hm_out && !need_vm && !h->multinomial();
        h->hm_is_neg = neg_only;
        launch_up(h, h->vs.p, h->vs.ld, B, last_out ? hm_out : (neg_only ? nullptr : h->hm.p), last ? nullptr : h->hs.p,
                  last_out ? h->H : h->hm.ld, h->cfg.sample_h_states, SITE_H, t, (!hm_out && last) ? h->hneg.p : nullptr);
        hstate = h->hs.p;
    }
    BM_TRY(chain_end(h));
    return 0;
}

static void fill_bias(bm_rbm *h, float N, float lr, float mom, RbmBiasArgs &b) {
    float *tail = h->grad.p + h->grad_tail();
    b.sv = tail; b.sh = tail + h->V; b.sq = tail + h->V + h->H;
    b.vb = h->vb.p; b.dvb = h->dvb.p; b.hb = h->hb.p; b.dhb = h->dhb.p; b.q = h->q.p; b.pen = h->pen.p;
    b.V = h->V; b.H = h->H;
    b.N = N; b.lr = lr; b.mom = mom;
    b.damping = h->cfg.sparsity_damping; b.cost = h->cfg.sparsity_cost; b.target = h->cfg.sparsity_target;
}

// single-GPU path: column sums + bias/q update in ONE launch (or inside the grad launch)
static int fill_bias_fused(bm_rbm *h, int B, float lr, float mom, RbmBiasFusedArgs &a) {
    memset(&a, 0, sizeof(a));
    a.X = h->Xin; a.ldx = h->Xin_ld; a.vs = h->vs.p; a.ldv = h->vs.ld;
    a.h0m = h->h0m.p; a.ldh0 = h->h0m.ld; a.B = B;
    // the last up-pass of a plain update leaves only -h_k behind (run_chain)
    if (h->hm_is_neg) { a.hm = h->hneg.p; a.ldh = h->hneg.ld; a.hm_negated = 1; }
    else              { a.hm = h->hm.p; a.ldh = h->hm.ld; }
    a.raw_tail = h->grad.p + h->grad_tail();
    RbmBiasArgs &b = a.u;
    b.vb = h->vb.p; b.dvb = h->dvb.p; b.hb = h->hb.p; b.dhb = h->dhb.p; b.q = h->q.p; b.pen = h->pen.p;
    b.V = h->V; b.H = h->H;
    b.N = (float)B; b.lr = lr; b.mom = mom;
    b.damping = h->cfg.sparsity_damping; b.cost = h->cfg.sparsity_cost; b.target = h->cfg.sparsity_target;
    return (h->V + 63) / 64 + (h->H + 63) / 64;
}

static void launch_bias_fused(bm_rbm *h, int B, float lr, float mom) {
    ProfScope _ps(h, KC_COLSUM);
    RbmBiasFusedArgs a;
    const int nw = fill_bias_fused(h, B, lr, mom, a);
    hipLaunchKernelGGL(rbm_bias_fused_kernel, dim3(nw), dim3(NT), 0, h->stream, a);
}

static void rbm_grad(bm_rbm *h, int B, int fused, float N, float lr, float mom, bool with_bias);

// whole parameter update of the fused single-GPU step (base_rbm.py:443-478)
static void launch_update_fused(bm_rbm *h, int B, float lr, float mom) {
    if (h->cfg.sparsity_cost != 0.f) {      // W update needs the penalty: bias kernel first
        launch_bias_fused(h, B, lr, mom);
        rbm_grad(h, B, 1, (float)B, lr, mom, false);
    } else {                                // penalty == 0: run both in one launch
        rbm_grad(h, B, 1, (float)B, lr, mom, true);
    }
}

static void fill_grad(bm_rbm *h, int B, int fused, float N, float lr, float mom, GradArgs &g) {
    memset(&g, 0, sizeof(g));
    g.Ppos = make_operand(h->h0m.p, h->h0m.ld, h->H);   // h0 means [k=b][i=h]           :447
    g.Qpos = make_operand(h->Xin, h->Xin_ld, h->V);     // X        [k=b][j=v]
    g.Kpos = B;
    g.Pneg = make_operand(h->hneg.p, h->hneg.ld, h->H); // -(h_k means): the chain subtracts  :448
    g.Qneg = make_operand(h->vs.p, h->vs.ld, h->V);     // v_k states
    g.Kneg = B;
    g.I = h->H; g.J = h->V;
    g.form = 0; g.fused = fused;
    g.raw = h->grad.p; g.raw2 = nullptr;
    g.W = h->W.p; g.dW = h->dW.p; g.Wt = nullptr;
    g.ldw = h->W.ld; g.ldwt = 0;
    if (h->use_wt && fused) { g.Wt = h->Wt.p; g.ldwt = h->Wt.ld; }
    g.N = N; g.M = N; g.l2 = h->cfg.l2; g.lr = lr; g.mom = mom;
}
static void rbm_grad(bm_rbm *h, int B, int fused, float N, float lr, float mom, bool with_bias) {
    ProfScope _ps(h, KC_GRAD);
    GradArgs g;
    fill_grad(h, B, fused, N, lr, mom, g);
    g.pen = with_bias ? nullptr : h->pen.p;
    if (with_bias) {
        g.nbias = fill_bias_fused(h, B, lr, mom, g.bias);
        g.bias.raw_only = fused ? 0 : 1;      // split (data-parallel) step: raw column sums only
    }
    bm::launch_grad(g, h->stream);
    if (fused && h->use_wt) h->wt_valid = true;      // the fused update wrote W and W^T together (every other writer of W
                                                     // clears the flag: set_param, apply_step, the exchange)
}

static void launch_fe(bm_rbm *h, const float *Xin, int ldx, int B, bool with_flip) {
    FeArgs f;
    memset(&f, 0, sizeof(f));
    f.P = make_operand(h->W.p, h->W.ld, h->H);
    f.Q = make_operand(Xin, ldx, B);
    f.K = h->V; f.I = h->H; f.J = B;
    f.hb = h->hb.p;
    f.rowacc = h->rowacc.p;
    if (with_flip) { f.rowacc2 = h->rowacc.p + h->maxB; f.flip_col = h->flip; }
    if (h->multinomial()) {                    // rbm.py:52-62: fresh h_hat draws, streams t = 0, 1, 2
        (void)hipMemsetAsync(h->hhat.p, 0, 3 * (size_t)h->H * sizeof(float), h->stream);
        const int M = h->cfg.n_samples;
        hipLaunchKernelGGL(mn_hhat_kernel, dim3((M + 255) / 256), dim3(256), 0, h->stream, h->hhat.p, h->H, M,
                           make_key(h, SITE_FE, 0), make_key(h, SITE_FE, 1), make_key(h, SITE_FE, 2));
        f.hvec = h->hhat.p;
        f.rowacc3 = h->rowacc.p + 2 * (size_t)h->maxB;
    }
    launch_fe_hidden(f, h->stream);
    FeRowArgs r;
    memset(&r, 0, sizeof(r));
    r.X = Xin; r.ld = ldx; r.V = h->V; r.B = B;
    r.vb = h->vb.p; r.sigma = (h->cfg.v_unit == BM_UNIT_GAUSSIAN) ? h->sigma.p : nullptr;
    r.rowacc = f.rowacc; r.rowacc2 = f.rowacc2; r.rowacc3 = f.rowacc3; r.flip_col = f.flip_col; r.out = h->scal + 2;
    hipLaunchKernelGGL(fe_row_kernel, dim3((B + FE_ROWS_PER_WG - 1) / FE_ROWS_PER_WG), dim3(256), 0, h->stream, r);
}

// -lgamma(M + K) + lgamma(M + 1) + lgamma(K)  (rbm.py:61); 0 for the other RBMs
static double mn_fe_const(const bm_rbm *h) {
    if (!h->multinomial()) return 0.0;
    const double M = h->cfg.n_samples, K = h->H;
    return -lgamma(M + K) + lgamma(M + 1.0) + lgamma(K);
}

// metrics from the chain currently in the handle (base_rbm.py:482-517)
static void metrics_to_out4(const bm_rbm *h, const double *host, int B, float *out4);
// The h0 pass can carry the free-energy sums when its pre-activation IS the free energy's (no dbm_first doubling), the hidden
// units are Bernoulli and the epilogue in use is act_kernel's (not the fast-binary strip kernel)
static bool metrics_fused_ok(const bm_rbm *h) {
    static const bool off = bm::dbg("metrics_fused") && atoi(bm::dbg("metrics_fused")) == 0;
    return !off && !h->multinomial() && !h->cfg.dbm_first && !h->fast_now && (h->W.ld & 3) == 0;
}
static void metrics_prep(bm_rbm *h, int B) {
    MetricsPrepArgs mp;
    mp.scal = h->scal; mp.rowacc = h->rowacc.p; mp.n_rowacc = 3 * h->maxB; mp.flip = h->flip; mp.B = B; mp.V = h->V;
    mp.key = make_key(h, SITE_PLL, 0); mp.row0 = (unsigned long long)h->row0;
    hipLaunchKernelGGL(metrics_prep_kernel, dim3(8), dim3(256), 0, h->stream, mp);
}
// the rest of the fetch, behind a run_chain(..., fetch = true): squared sums, the visible terms of the free energies
static int metrics_from_chain(bm_rbm *h, int B, float *out4) {
    const SqJob msre{h->Xin, h->Xin_ld, h->vm.p, h->vm.ld, B, h->V, h->scal + 0};                        // :486-488
    const SqJob l2{h->W.p, h->W.ld, nullptr, 0, h->V, h->H, h->scal + 1};                                 // :482-484
    if (h->fe_in_chain) {       // the hidden terms are in fe_part already: squared sums and the row sums as ONE launch
        FeRowArgs r;
        memset(&r, 0, sizeof(r));
        r.X = h->Xin; r.ld = h->Xin_ld; r.V = h->V; r.B = B;
        r.vb = h->vb.p; r.sigma = (h->cfg.v_unit == BM_UNIT_GAUSSIAN) ? h->sigma.p : nullptr;
        r.nslot = (h->H + 15) / 16; r.ld_part = (r.nslot + 3) & ~3;
        r.rowacc = h->fe_part.p; r.rowacc2 = h->fe_part.p + (size_t)r.ld_part * h->maxB; r.out = h->scal + 2;
        r.flip_col = nullptr; r.has_key = 1; r.key = make_key(h, SITE_PLL, 0); r.row0 = (unsigned long long)h->row0;
        const int nb_sq = 256, nb_fe = (B + FE_ROWS_PER_WG - 1) / FE_ROWS_PER_WG;
        hipLaunchKernelGGL(metrics_tail_kernel, dim3(nb_sq + nb_fe), dim3(256), 0, h->stream, msre, l2, r, nb_sq);
    } else {
        hipLaunchKernelGGL(sqdiff2_kernel, dim3(256), dim3(256), 0, h->stream, msre, l2);
        launch_fe(h, h->Xin, h->Xin_ld, B, true);
    }
    if (!out4) return 0;                                    // asynchronous caller: the sums stay in h->scal
    double host[6];
    BM_HIP(hipMemcpyAsync(host, h->scal, sizeof(host), hipMemcpyDeviceToHost, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
    metrics_to_out4(h, host, B, out4);
    return 0;
}
static void metrics_to_out4(const bm_rbm *h, const double *host, int B, float *out4) {
    // MultinomialRBM: every _free_energy() call draws its own h_hat (host[4] = F(x) of the PLL pair)
    // and carries the constant of rbm.py:61 (it cancels in the PLL difference)
    const double fe = host[2] / B + mn_fe_const(h), fe1 = h->multinomial() ? host[4] / B + mn_fe_const(h) : fe;
    const double fe2 = host[3] / B + mn_fe_const(h);
    out4[0] = (float)(host[0] / ((double)B * h->V));            // msre          :487
    const float d = (float)(fe2 - fe1);                         // pll           :511-512
    const float ls = -(fmaxf(-d, 0.f) + log1pf(expf(-fabsf(d))));
    out4[1] = (float)h->V * ls;
    out4[2] = h->cfg.l2 * (float)(0.5 * host[1]);               // l2_loss       :483
    out4[3] = (float)fe;                                        // free energy   :516
}

extern "C" {

const char *bm_last_error(void) { return bm::g_err; }
const char *bm_version(void) { return "bm355 0.1 gfx950"; }

int bm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
// Host waits (bm_*_sync, the blocking reads of metrics) SPIN instead of sleeping on an interrupt: a training loop waits
// for the device every few hundred microseconds (metric fetches, the mean-field loop control), and the wake-up of a
// blocked host thread costs tens of microseconds per wait.  BM355_HOST_WAIT=yield|block selects the other policies.
// The flag can only be set before the device's context exists: when the host framework created it first (torch), the
// call fails harmlessly and the framework's policy stays.
int bm_set_device(int device) {
    BM_HIP(hipSetDevice(device));
    const char *w = getenv("BM355_HOST_WAIT");
    unsigned flags = hipDeviceScheduleSpin;
    if (w && !strcmp(w, "yield")) flags = hipDeviceScheduleYield;
    if (w && !strcmp(w, "block")) flags = hipDeviceScheduleBlockingSync;
    if (hipSetDeviceFlags(flags) != hipSuccess) (void)hipGetLastError();
    return 0;
}
int bm_dev_alloc(size_t bytes, void **out_dev) { BM_HIP(hipMalloc(out_dev, bytes ? bytes : 1)); return 0; }
int bm_dev_free(void *dev) { BM_HIP(hipFree(dev)); return 0; }
// Host -> device.  Large pageable sources (a training set) are pinned in place for the duration of the copy: the
// runtime otherwise stages them through its own bounce buffers at a fraction of the link rate.  BM355_DEBUG=h2d_pin=0
// keeps the plain copy; any failure of the registration falls back to it.
int bm_h2d(void *dst, const void *src, size_t bytes) {
    static const bool pin = !(bm::dbg("h2d_pin") && atoi(bm::dbg("h2d_pin")) == 0);
    if (pin && bytes >= ((size_t)32 << 20)) {
        if (hipHostRegister(const_cast<void *>(src), bytes, hipHostRegisterDefault) == hipSuccess) {
            const hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
            (void)hipHostUnregister(const_cast<void *>(src));
            BM_HIP(e);
            return 0;
        }
        (void)hipGetLastError();
    }
    BM_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
int bm_d2h(void *dst, const void *src, size_t bytes) { BM_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return 0; }
int bm_dev_memset(void *dst, int value, size_t bytes) { BM_HIP(hipMemset(dst, value, bytes)); return 0; }

int bm_rbm_create(const bm_rbm_config *cfg, bm_rbm **out) {
    BM_CHECK(cfg && out, "null argument");
    BM_CHECK(cfg->n_visible >= 1 && cfg->n_hidden >= 1, "bad layer sizes %d x %d", cfg->n_visible, cfg->n_hidden);
    BM_CHECK(cfg->max_batch >= 1, "max_batch must be >= 1");
    BM_CHECK(cfg->v_unit == BM_UNIT_BERNOULLI || cfg->v_unit == BM_UNIT_GAUSSIAN, "unknown visible unit %d", cfg->v_unit);
    BM_CHECK(cfg->h_unit == BM_UNIT_BERNOULLI || cfg->h_unit == BM_UNIT_MULTINOMIAL, "unknown hidden unit %d", cfg->h_unit);
    if (cfg->h_unit == BM_UNIT_MULTINOMIAL) {
        BM_CHECK(cfg->n_samples >= 1, "MultinomialRBM: n_samples must be >= 1 (got %d)", cfg->n_samples);
        BM_CHECK(cfg->n_hidden <= 8192, "MultinomialRBM: n_hidden %d > 8192 (softmax row staged in LDS)", cfg->n_hidden);
    }
    BM_CHECK(bm_device_count() > 0, "no HIP device visible: libbm355 has no CPU fallback");
    bm_rbm *h = new bm_rbm();
    h->cfg = *cfg;
    h->V = cfg->n_visible; h->H = cfg->n_hidden; h->maxB = cfg->max_batch;
    const int V = h->V, H = h->H, B = h->maxB;
    BM_HIP(hipGetDevice(&h->device));
    BM_HIP(hipStreamCreate(&h->stream));
    BM_HIP(hipEventCreate(&h->ev0));
    BM_HIP(hipEventCreate(&h->ev1));
    BM_TRY(h->W.alloc(V, H)); BM_TRY(h->dW.alloc(V, H));
    {
        const char *e = bm::dbg("up_xm");          // default on; 0 keeps the k-major prop-up
        h->use_wt = !(e && atoi(e) == 0) && (V % 4 == 0) && !h->multinomial();
        if (h->use_wt) BM_TRY(h->Wt.alloc(H, V));
    }
    BM_TRY(h->vb.alloc(V)); BM_TRY(h->hb.alloc(H)); BM_TRY(h->dvb.alloc(V)); BM_TRY(h->dhb.alloc(H));
    BM_TRY(h->q.alloc(H)); BM_TRY(h->sigma.alloc(V));
    BM_TRY(h->h0m.alloc(B, H)); BM_TRY(h->h0s.alloc(B, H)); BM_TRY(h->hm.alloc(B, H)); BM_TRY(h->hs.alloc(B, H)); BM_TRY(h->hneg.alloc(B, H));
    BM_TRY(h->vm.alloc(B, V)); BM_TRY(h->vs.alloc(B, V)); BM_TRY(h->Xs.alloc(B, V)); BM_TRY(h->Xd.alloc(B, V));
    BM_TRY(h->grad.alloc(h->grad_tail() + V + 2 * (size_t)H));
    BM_TRY(h->pen.alloc(H));
    BM_TRY(h->rowacc.alloc(3 * (size_t)B)); BM_TRY(h->hhat.alloc(3 * (size_t)H));
    BM_HIP(hipMalloc((void **)&h->flip, B * sizeof(int)));
    BM_HIP(hipMalloc((void **)&h->scal, 6 * sizeof(double)));
    {   // sigma defaults to 1 (rbm.py:88)
        std::vector<float> ones(V, 1.0f);
        BM_HIP(hipMemcpy(h->sigma.p, ones.data(), V * sizeof(float), hipMemcpyHostToDevice));
    }
    *out = h;
    return 0;
}

int bm_rbm_destroy(bm_rbm *h) {
    if (!h) return 0;
    (void)hipStreamSynchronize(h->stream);
    if (h->xchg_used) xchg_bind_user(h->xchg_used, nullptr);
    Mat *mats[] = {&h->W, &h->dW, &h->Wt, &h->h0m, &h->h0s, &h->hm, &h->hs, &h->hneg, &h->vm, &h->vs, &h->Xs, &h->Xd};
    for (Mat *m : mats) m->release();
    DevBuf *all[] = {&h->vb, &h->hb, &h->dvb, &h->dhb, &h->q, &h->sigma, &h->grad, &h->grad_alt, &h->pen, &h->rowacc, &h->hhat, &h->fe_part};
    for (int i = 0; i < 2; ++i) {
        if (h->ev_ready[i]) (void)hipEventDestroy(h->ev_ready[i]);
        if (h->ev_reduced[i]) (void)hipEventDestroy(h->ev_reduced[i]);
    }
    if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
    if (h->stage_stream) { (void)hipStreamSynchronize(h->stage_stream); (void)hipStreamDestroy(h->stage_stream); }
    if (h->stage_host) (void)hipHostFree(h->stage_host);
    if (h->mring) (void)hipHostFree(h->mring);
    if (h->ev_mlast) (void)hipEventDestroy(h->ev_mlast);
    for (auto &sg : h->stage) {
        sg.W.release(); sg.dW.release();
        DevBuf *sv[] = {&sg.vb, &sg.hb, &sg.dvb, &sg.dhb, &sg.q, &sg.sigma};
        for (DevBuf *b : sv) b->release();
        if (sg.ev) (void)hipEventDestroy(sg.ev);
    }
    for (DevBuf *b : all) b->release();
    h->W3.release(); h->W3t.release(); h->hs16.release(); h->vs16.release();
    if (h->nonbinary) (void)hipFree(h->nonbinary);
    if (h->flip) (void)hipFree(h->flip);
    if (h->scal) (void)hipFree(h->scal);
    h->chain.release();
    for (auto &r : h->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    (void)hipEventDestroy(h->ev0);
    (void)hipEventDestroy(h->ev1);
    (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

// Status words the device leaves behind (the stream `s` is idle up to the point of interest when this is called):
// the direct exchange's time-out word and the chained launches' (bm_chain.h).  A failed chained launch is reported
// ONCE - the results since the last check are invalid - and switches chaining off for this handle: later passes run as
// per-pass launches instead of leaving the handle poisoned (round-4 advisor).
static int check_device_status(bm_rbm *h) {
    if (h->xchg_used) BM_TRY(xchg_check_status(h->xchg_used));      // a lost rank is an ERROR here, never a silent wrong sum
    if (h->chain.status) {      // chained launches (bm_chain.h): an expired wait or a tile nobody computed is an ERROR
        int st[2] = {0, 0};
        BM_HIP(hipMemcpy(st, h->chain.status, sizeof(st), hipMemcpyDeviceToHost));
        const long long expect = h->chain.tiles_expected & 0xffffffffLL;
        if (st[0] != 0 || (long long)(unsigned)st[1] != expect) {
            h->chain.mode = 0;                                   // per-pass launches from here on
            h->chain.tiles_expected = 0;
            BM_HIP(hipMemset(h->chain.status, 0, sizeof(st)));
            BM_CHECK(st[0] == 0, "chained propagation launch failed (status %d: %s); the results since the last check are "
                     "invalid, chained launches are now off for this handle", st[0],
                     st[0] == CHAIN_ERR_TIMEOUT ? "a wait for a producing tile expired" : "not an 8-XCD device");
            BM_CHECK(false, "chained propagation launches computed %u tiles, expected %lld; the results since the last check "
                     "are invalid, chained launches are now off for this handle", (unsigned)st[1], expect);
        }
    }
    return 0;
}

int bm_rbm_sync(bm_rbm *h) {
    BM_HIP(hipStreamSynchronize(h->stream));
    BM_TRY(check_device_status(h));
    if (h->nonbinary) {
        int bad = 0;
        BM_HIP(hipMemcpy(&bad, h->nonbinary, sizeof(int), hipMemcpyDeviceToHost));
        if (bad) {
            BM_HIP(hipMemset(h->nonbinary, 0, sizeof(int)));
            BM_CHECK(false, "fast-binary mode: bm_rbm_gibbs was given hidden states that are not a {0,1} bitmap");
        }
    }
    return 0;
}

// Opt-in fast-binary mode (bm_bf3.h): the sampling sweep (bm_rbm_gibbs) of a Bernoulli-Bernoulli RBM with both layers
// sampled runs its contractions as exact-product bf16 x 3 on the bf16 matrix cores; agreement with the default path
// is to fp32 round-off, not bit for bit.  0 restores the default.
// 1 = where it PAYS: at 784 x 1024 x 512 the bf16 strip kernel is slower than the fp32 path (25.9 against 25.2 us per sweep,
// profiles/r5_gibbs_summary.md: the pass is bound by its fill and epilogue, not by matrix time), so the switch only takes
// effect from 8M weights upwards (the 3072 x 5000 shape gains); 2 = wherever legal (tests, measurements).
int bm_rbm_set_fast_binary(bm_rbm *h, int32_t on) {
    BM_CHECK(h, "null argument");
    h->fast = (on >= 2 || (on == 1 && (long long)h->V * h->H >= (8ll << 20))) ? 1 : 0;
    return 0;
}

// name -> vector variable
static DevBuf *find_vec(bm_rbm *h, const std::string &n) {
    if (n == "vb") return &h->vb;
    if (n == "hb") return &h->hb;
    if (n == "dvb") return &h->dvb;
    if (n == "dhb") return &h->dhb;
    if (n == "q_means") return &h->q;
    if (n == "sigma") return &h->sigma;
    return nullptr;
}

int bm_rbm_set_param(bm_rbm *h, const char *name, const float *host, size_t n) {
    const std::string nm(name ? name : "");
    BM_HIP(hipStreamSynchronize(h->stream));
    if (nm == "W" || nm == "dW") {
        BM_CHECK(n == (size_t)h->V * h->H, "variable '%s' has %zu elements, got %zu", name, (size_t)h->V * h->H, n);
        BM_TRY((nm == "W" ? h->W : h->dW).upload(host));
        if (nm == "W") h->wt_valid = false;
        else { h->dw_sharded = false; if (h->xchg_used) xchg_dw_replaced(h->xchg_used); }
        return 0;
    }
    DevBuf *b = find_vec(h, nm);
    BM_CHECK(b, "unknown RBM variable '%s'", name ? name : "(null)");
    BM_CHECK(n == b->n, "variable '%s' has %zu elements, got %zu", name, b->n, n);
    BM_HIP(hipMemcpy(b->p, host, n * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

// set a variable from DEVICE memory (dense, row-major), asynchronously, in stream order - no host synchronisation:
// re-initialising a model between runs, or `init_from` another handle's variables, without idling the GPU
int bm_rbm_set_param_dev(bm_rbm *h, const char *name, const float *src_dev, size_t n) {
    const std::string nm(name ? name : "");
    BM_CHECK(h && src_dev, "null argument");
    if (nm == "W" || nm == "dW") {
        BM_CHECK(n == (size_t)h->V * h->H, "variable '%s' has %zu elements, got %zu", name, (size_t)h->V * h->H, n);
        Mat &m = nm == "W" ? h->W : h->dW;
        if (nm == "W") h->wt_valid = false;
        else { h->dw_sharded = false; if (h->xchg_used) xchg_dw_replaced(h->xchg_used); }
        BM_HIP(hipMemcpy2DAsync(m.p, (size_t)m.ld * sizeof(float), src_dev, (size_t)m.cols * sizeof(float),
                                (size_t)m.cols * sizeof(float), m.rows, hipMemcpyDeviceToDevice, h->stream));
        return 0;
    }
    DevBuf *b = find_vec(h, nm);
    BM_CHECK(b, "unknown RBM variable '%s'", name ? name : "(null)");
    BM_CHECK(n == b->n, "variable '%s' has %zu elements, got %zu", name, b->n, n);
    BM_HIP(hipMemcpyAsync(b->p, src_dev, n * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    return 0;
}

int bm_rbm_get_param(bm_rbm *h, const char *name, float *host, size_t n) {
    const std::string nm(name ? name : "");
    BM_HIP(hipStreamSynchronize(h->stream));
    BM_TRY(check_device_status(h));        // never hand out variables computed from invalid tiles / a lost rank's sums
    if (nm == "W" || nm == "dW") {
        BM_CHECK(n == (size_t)h->V * h->H, "variable '%s' has %zu elements, got %zu", name, (size_t)h->V * h->H, n);
        if (nm == "dW") BM_TRY(check_dw(h, "bm_rbm_get_param(dW)"));
        return (nm == "W" ? h->W : h->dW).download(host);
    }
    DevBuf *b = find_vec(h, nm);
    BM_CHECK(b, "unknown RBM variable '%s'", name ? name : "(null)");
    BM_CHECK(n == b->n, "variable '%s' has %zu elements, got %zu", name, b->n, n);
    BM_HIP(hipMemcpy(host, b->p, n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// Snapshot without a host wait (the checkpoint of an epoch while the next one already runs): bm_rbm_stage copies
// every variable into slot `slot` (0 | 1) device-to-device in stream order and returns at once; bm_rbm_get_staged
// reads a variable of that snapshot back on its own stream (it waits for the copies only) and may be called from
// another host thread while the engine's stream keeps working.  The caller must not re-stage a slot that is still
// being read.
int bm_rbm_stage(bm_rbm *h, int32_t slot) {
    BM_CHECK(h && (slot == 0 || slot == 1), "bad stage slot %d", (int)slot);
    bm_rbm::Stage &sg = h->stage[slot];
    // a slot that another thread is reading back must not be overwritten under it (any caller of the C API, not
    // only the Python bookkeeping of base.py: round-3 advisor)
    BM_CHECK(sg.readers.load() == 0, "stage slot %d is being read by bm_rbm_get_staged: use the other slot", (int)slot);
    BM_TRY(check_dw(h, "bm_rbm_stage"));
    if (!sg.ev) {
        BM_TRY(sg.W.alloc(h->V, h->H)); BM_TRY(sg.dW.alloc(h->V, h->H));
        BM_TRY(sg.vb.alloc(h->V)); BM_TRY(sg.dvb.alloc(h->V)); BM_TRY(sg.sigma.alloc(h->V));
        BM_TRY(sg.hb.alloc(h->H)); BM_TRY(sg.dhb.alloc(h->H)); BM_TRY(sg.q.alloc(h->H));
        BM_HIP(hipEventCreateWithFlags(&sg.ev, hipEventDisableTiming));
    }
    if (!h->stage_stream) BM_HIP(hipStreamCreateWithFlags(&h->stage_stream, hipStreamNonBlocking));
    // Bound the host's run-ahead to one snapshot interval: a training loop that never fetches anything would otherwise
    // queue every epoch of the call at once (measured: 4000 updates = 16 000 launches in flight ran 101 instead of
    // 77 us per update).  Waiting for the PREVIOUS snapshot's copies leaves the whole current epoch queued: no bubble.
    static const bool bound = !(bm::dbg("stage_ahead") && atoi(bm::dbg("stage_ahead")) == 0);
    if (bound && h->last_stage >= 0) BM_HIP(hipEventSynchronize(h->stage[h->last_stage].ev));
    BM_HIP(hipMemcpyAsync(sg.W.p, h->W.p, h->W.count() * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    BM_HIP(hipMemcpyAsync(sg.dW.p, h->dW.p, h->dW.count() * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    DevBuf *src[] = {&h->vb, &h->hb, &h->dvb, &h->dhb, &h->q, &h->sigma};
    DevBuf *dst[] = {&sg.vb, &sg.hb, &sg.dvb, &sg.dhb, &sg.q, &sg.sigma};
    for (int i = 0; i < 6; ++i)
        BM_HIP(hipMemcpyAsync(dst[i]->p, src[i]->p, src[i]->n * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    BM_HIP(hipEventRecord(sg.ev, h->stream));
    sg.ready = true;
    h->last_stage = slot;
    return 0;
}
int bm_rbm_get_staged(bm_rbm *h, int32_t slot, const char *name, float *host, size_t n) {
    BM_CHECK(h && (slot == 0 || slot == 1) && h->stage[slot].ready, "stage slot %d holds no snapshot", (int)slot);
    BM_HIP(hipSetDevice(h->device));                 // (the calling thread may be a fresh one)
    bm_rbm::Stage &sg = h->stage[slot];
    struct Reading { std::atomic<int> &n; Reading(std::atomic<int> &r) : n(r) { ++n; } ~Reading() { --n; } } reading(sg.readers);
    const std::string nm(name ? name : "");
    // Wait for the staged copies on the HOST first, then copy through a pinned bounce buffer: a pageable
    // device-to-host copy that has to wait for an event parks inside the runtime (measured: the training thread's
    // launches stalled behind it, 101 instead of 76 us per update)
    BM_HIP(hipEventSynchronize(sg.ev));
    const size_t need = (size_t)h->V * h->H * sizeof(float);
    if (!h->stage_host) BM_HIP(hipHostMalloc((void **)&h->stage_host, need, hipHostMallocDefault));
    if (h->chain.status) {
        // the sticky error word of the chained launches (not the tile count: the training thread may be ahead of the
        // device): a checkpoint must not be written from tiles a failed launch left invalid
        BM_HIP(hipMemcpyAsync(h->stage_host, h->chain.status, sizeof(int), hipMemcpyDeviceToHost, h->stage_stream));
        BM_HIP(hipStreamSynchronize(h->stage_stream));
        BM_CHECK(*(const int *)h->stage_host == 0, "a chained propagation launch failed (status %d) before this snapshot: "
                 "it is invalid (bm_rbm_sync reports and recovers)", *(const int *)h->stage_host);
    }
    if (nm == "W" || nm == "dW") {
        BM_CHECK(n == (size_t)h->V * h->H, "variable '%s' has %zu elements, got %zu", name, (size_t)h->V * h->H, n);
        const Mat &m = nm == "W" ? sg.W : sg.dW;
        BM_HIP(hipMemcpy2DAsync(h->stage_host, (size_t)m.cols * sizeof(float), m.p, (size_t)m.ld * sizeof(float),
                                (size_t)m.cols * sizeof(float), m.rows, hipMemcpyDeviceToHost, h->stage_stream));
    } else {
        const char *names[] = {"vb", "hb", "dvb", "dhb", "q_means", "sigma"};
        DevBuf *bufs[] = {&sg.vb, &sg.hb, &sg.dvb, &sg.dhb, &sg.q, &sg.sigma};
        DevBuf *b = nullptr;
        for (int i = 0; i < 6; ++i) if (nm == names[i]) b = bufs[i];
        BM_CHECK(b, "unknown RBM variable '%s'", name ? name : "(null)");
        BM_CHECK(n == b->n, "variable '%s' has %zu elements, got %zu", name, b->n, n);
        BM_CHECK(n * sizeof(float) <= need, "variable '%s' larger than the bounce buffer", name);
        BM_HIP(hipMemcpyAsync(h->stage_host, b->p, n * sizeof(float), hipMemcpyDeviceToHost, h->stage_stream));
    }
    BM_HIP(hipStreamSynchronize(h->stage_stream));
    memcpy(host, h->stage_host, n * sizeof(float));
    return 0;
}

int bm_rbm_dev_ptr(bm_rbm *h, const char *name, void **out_dev, size_t *out_n) {
    const std::string nm(name ? name : "");
    DevBuf *b = (nm == "grad") ? &h->grad : find_vec(h, nm);
    BM_CHECK(b, "no device view for '%s' (matrices are pitched; use get/set_param)", name ? name : "(null)");
    *out_dev = b->p;
    if (out_n) *out_n = b->n;
    return 0;
}

int bm_rbm_seed(bm_rbm *h, uint64_t seed) { h->seed = seed; h->call = 0; return 0; }
int bm_rbm_set_row_offset(bm_rbm *h, int64_t row0) { h->row0 = row0; return 0; }

int bm_rbm_train_step(bm_rbm *h, const float *X_dev, int32_t B, float lr, float mom, int32_t k) {
    BM_TRY(check_dw(h, "bm_rbm_train_step"));
    BM_TRY(run_chain(h, X_dev, B, k, nullptr, false, true));
    launch_update_fused(h, B, lr, mom);
    h->call++;
    BM_HIP(hipGetLastError());
    return 0;
}

int bm_rbm_train_step_metrics(bm_rbm *h, const float *X_dev, int32_t B, float lr, float mom, int32_t k,
                              float *out4) {
    BM_TRY(check_dw(h, "bm_rbm_train_step_metrics"));
    BM_TRY(run_chain(h, X_dev, B, k, nullptr, true, false, false, true));
    BM_TRY(metrics_from_chain(h, B, out4));
    launch_update_fused(h, B, lr, mom);
    h->call++;
    BM_HIP(hipGetLastError());
    return 0;
}

// The same fetch WITHOUT the host wait: the reference reads the train metrics every `train_metrics_every_iter`-th
// iteration but only uses their mean at the end of the epoch (base_rbm.py:549-571), so the six device sums of a metrics
// iteration are copied into a pinned ring in stream order and converted when bm_rbm_collect_metrics is called (one
// synchronisation per epoch instead of one per fetch: fit() with the reference's default cadence ran 81 - 98 us per
// update against 68 without metrics, almost all of it the GPU idling behind the host round trips).
int bm_rbm_train_step_metrics_async(bm_rbm *h, const float *X_dev, int32_t B, float lr, float mom, int32_t k) {
    if (!h->mring) {
        BM_HIP(hipHostMalloc((void **)&h->mring, (size_t)bm_rbm::MRING * 6 * sizeof(double), hipHostMallocDefault));
        BM_HIP(hipHostGetDevicePointer((void **)&h->mring_dev, h->mring, 0));
        h->mring_B.resize(bm_rbm::MRING);
    }
    BM_CHECK(h->mring_n < bm_rbm::MRING, "%d metric fetches are pending: call bm_rbm_collect_metrics", h->mring_n);
    BM_TRY(check_dw(h, "bm_rbm_train_step_metrics_async"));
    BM_TRY(run_chain(h, X_dev, B, k, nullptr, true, false, false, true));
    BM_TRY(metrics_from_chain(h, B, nullptr));
    hipLaunchKernelGGL(scal_to_host_kernel, dim3(1), dim3(64), 0, h->stream, (const double *)h->scal,
                       h->mring_dev + (size_t)h->mring_n * 6);
    if (!h->ev_mlast) BM_HIP(hipEventCreateWithFlags(&h->ev_mlast, hipEventDisableTiming));
    BM_HIP(hipEventRecord(h->ev_mlast, h->stream));
    h->mring_B[h->mring_n++] = B;
    launch_update_fused(h, B, lr, mom);
    h->call++;
    BM_HIP(hipGetLastError());
    return 0;
}
// out4n [max_n][4] <- the pending fetches in order (msre, pll, l2_loss, free energy each); *out_n their number
int bm_rbm_collect_metrics(bm_rbm *h, float *out4n, int32_t max_n, int32_t *out_n) {
    BM_CHECK(h && out_n && (out4n || max_n == 0), "null argument");
    BM_CHECK(h->mring_n <= max_n, "%d fetches pending, room for %d", h->mring_n, (int)max_n);
    // wait for the last FETCH, not for the stream: what was queued behind it keeps running while the host reads the ring
    if (h->mring_n > 0 && h->ev_mlast) BM_HIP(hipEventSynchronize(h->ev_mlast));
    if (h->chain.status || h->xchg_used) {
        // (status words are read with a blocking copy: that waits for the whole stream - only handles that use chained
        //  launches or a direct exchange pay it)
        BM_HIP(hipStreamSynchronize(h->stream));
        // the pending fetches are dropped either way: an error must not leave them for the next epoch's mean
        const int rc = check_device_status(h);
        if (rc) { h->mring_n = 0; *out_n = 0; return rc; }
    }
    for (int i = 0; i < h->mring_n; ++i) metrics_to_out4(h, h->mring + (size_t)i * 6, h->mring_B[i], out4n + 4 * (size_t)i);
    *out_n = h->mring_n;
    h->mring_n = 0;
    return 0;
}

// N rows of X_dev as consecutive minibatches of `batch` rows, driven from C: the same launches and RNG call counters
// as the caller's own loop over bm_rbm_train_step, without a host round trip per batch.  (Round 3 could replay recurring
// runs of updates from a HIP graph, bit-exactly, and measured it SLOWER on MI355X / ROCm 7.2 - 66.3 against 65.4 us per
// update over 2000 updates, 75 - 77 against 69 us for a single 20-update replay; removed in round 4, profiles/NOTES.md 3.10.)
int bm_rbm_train_epoch(bm_rbm *h, const float *X_dev, int64_t N, int32_t batch, float lr, float mom, int32_t k) {
    BM_CHECK(batch >= 1 && N >= 1, "bad N=%lld batch=%d", (long long)N, batch);
    for (int64_t s = 0; s < N; s += batch) {
        const int B = (int)((N - s < batch) ? (N - s) : batch);
        BM_TRY(bm_rbm_train_step(h, X_dev + (size_t)s * h->V, B, lr, mom, k));
    }
    return 0;
}

int bm_rbm_grad_step(bm_rbm *h, const float *X_dev, int32_t B, int32_t k) {
    BM_TRY(run_chain(h, X_dev, B, k, nullptr, false, false, true));
    rbm_grad(h, B, 0, (float)B, 0.f, 0.f, true);     // raw outer products + raw column sums, one launch
    h->call++;
    BM_HIP(hipGetLastError());
    return 0;
}

// ---- delayed-gradient data parallelism (a documented NON-parity mode, SURVEY 8e / DESIGN 6): two gradient slots.
// Step t: grad_step into slot t % 2, its all-reduce goes out on the communication stream and runs under step t+1's
// Gibbs chain; the update applied at the end of step t is the (already reduced) one of step t-1.
static int ensure_delayed(bm_rbm *h) {
    if (h->comm_stream) return 0;
    BM_TRY(h->grad_alt.alloc(h->grad.n));
    BM_HIP(hipStreamCreate(&h->comm_stream));
    for (int i = 0; i < 2; ++i) {
        BM_HIP(hipEventCreateWithFlags(&h->ev_ready[i], hipEventDisableTiming));
        BM_HIP(hipEventCreateWithFlags(&h->ev_reduced[i], hipEventDisableTiming));
    }
    return 0;
}
int bm_rbm_set_grad_slot(bm_rbm *h, int32_t slot) {
    BM_CHECK(h && (slot == 0 || slot == 1), "slot must be 0 or 1");
    if (slot == h->grad_slot) return 0;
    BM_TRY(ensure_delayed(h));
    DevBuf t = h->grad; h->grad = h->grad_alt; h->grad_alt = t;
    h->grad_slot = slot;
    return 0;
}
int bm_rbm_allreduce_grads_async(bm_rbm *h, bm_comm *c) {
    BM_CHECK(h && c, "null argument");
    BM_TRY(ensure_delayed(h));
    const int s = h->grad_slot;
    BM_HIP(hipEventRecord(h->ev_ready[s], h->stream));
    BM_HIP(hipStreamWaitEvent(h->comm_stream, h->ev_ready[s], 0));
    BM_TRY(bm_comm_allreduce_sum(c, h->grad.p, h->grad.n, (void *)h->comm_stream));
    BM_HIP(hipEventRecord(h->ev_reduced[s], h->comm_stream));
    return 0;
}
int bm_rbm_wait_grads(bm_rbm *h, int32_t slot) {
    BM_CHECK(h && (slot == 0 || slot == 1) && h->comm_stream, "no reduction was started on slot %d", slot);
    BM_HIP(hipStreamWaitEvent(h->stream, h->ev_reduced[slot], 0));
    return 0;
}

int bm_rbm_apply_step(bm_rbm *h, int32_t B_global, float lr, float mom) {
    BM_TRY(check_dw(h, "bm_rbm_apply_step"));
    ProfScope _ps(h, KC_BIAS);
    RbmBiasArgs b;
    fill_bias(h, (float)B_global, lr, mom, b);
    ApplyWArgs a;
    memset(&a, 0, sizeof(a));
    a.raw = h->grad.p; a.raw2 = nullptr;
    a.W = h->W.p; a.dW = h->dW.p; a.Wt = nullptr;
    h->wt_valid = false;
    a.I = h->H; a.J = h->V; a.ldw = h->W.ld; a.ldwt = 0; a.form = 0;
    a.N = (float)B_global; a.M = a.N; a.l2 = h->cfg.l2; a.lr = lr; a.mom = mom;
    if (h->cfg.sparsity_cost != 0.f) {      // the W update needs the penalty: bias update first
        hipLaunchKernelGGL(rbm_bias_kernel, dim3((h->V + h->H + 255) / 256), dim3(256), 0, h->stream, b);
        a.pen = h->pen.p;
        launch_apply_w(a, nullptr, h->stream);
    } else {                                // penalty == 0 (pen stays 0): one launch for both
        a.pen = nullptr;
        launch_apply_w(a, &b, h->stream);
    }
    BM_HIP(hipGetLastError());
    return 0;
}

int bm_rbm_transform(bm_rbm *h, const float *X_dev, int32_t B, int32_t k, float *H_dev) {
    BM_CHECK(H_dev, "null output");
    BM_TRY(run_chain(h, X_dev, B, k, H_dev));
    h->call++;
    BM_HIP(hipGetLastError());
    return 0;
}

int bm_rbm_metrics(bm_rbm *h, const float *X_dev, int32_t B, int32_t k, float *out4) {
    BM_TRY(run_chain(h, X_dev, B, k, nullptr, true, false, false, true));
    BM_TRY(metrics_from_chain(h, B, out4));
    h->call++;
    return 0;
}

int bm_rbm_free_energy(bm_rbm *h, const float *X_dev, int32_t B, float *out1) {
    BM_CHECK(B >= 1 && B <= h->maxB, "batch %d outside [1, max_batch=%d]", B, h->maxB);
    const float *Xin = X_dev;
    int ldx = h->V;
    if (h->cfg.v_unit == BM_UNIT_GAUSSIAN) {
        hipLaunchKernelGGL(div_cols_kernel, dim3(256), dim3(256), 0, h->stream, Xin, ldx, h->sigma.p, h->Xs.p,
                           h->Xs.ld, B, h->V);
        Xin = h->Xs.p; ldx = h->Xs.ld;
    }
    bool dropped = false;
    if (h->cfg.dropout >= 0.f) {
        // free_energy_op is built from self._X_batch AFTER tf.nn.dropout replaced it (base_rbm.py:417-418,
        // :516): the `feg` metric sees the dropped input like msre / pll do
        hipLaunchKernelGGL(dropout_kernel, dim3(256), dim3(256), 0, h->stream, Xin, ldx, h->Xd.p, h->Xd.ld, B, h->V,
                           h->cfg.dropout, make_key(h, SITE_DROPOUT, 0),
                           (unsigned long long)h->row0 * (unsigned long long)h->V);
        Xin = h->Xd.p; ldx = h->Xd.ld;
        dropped = true;
    }
    BM_HIP(hipMemsetAsync(h->scal, 0, 6 * sizeof(double), h->stream));
    BM_HIP(hipMemsetAsync(h->rowacc.p, 0, 3 * (size_t)h->maxB * sizeof(float), h->stream));
    launch_fe(h, Xin, ldx, B, false);
    double host[6];
    BM_HIP(hipMemcpyAsync(host, h->scal, sizeof(host), hipMemcpyDeviceToHost, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
    *out1 = (float)(host[2] / B + mn_fe_const(h));
    if (h->multinomial() || dropped) h->call++;   // the random h_hat / the dropout mask consumed one call of the stream
    return 0;
}

int bm_rbm_gibbs(bm_rbm *h, float *H_dev, float *V_dev, int32_t B, int32_t n_steps) {
    BM_CHECK(B >= 1 && B <= h->maxB, "batch %d outside [1, max_batch=%d]", B, h->maxB);
    BM_CHECK(H_dev && V_dev, "null state pointer");
    BM_CHECK(n_steps >= 1, "n_steps must be >= 1");
    struct FastScope { bm_rbm *h; ~FastScope() { h->fast_now = false; } } fast_scope{h};
    const bool fast = h->fast && h->cfg.v_unit == BM_UNIT_BERNOULLI && !h->multinomial() && h->cfg.sample_v_states &&
                      h->cfg.sample_h_states && h->cfg.dropout < 0.f;
    if (!fast && !h->multinomial()) {
        // The sweeps read and write the caller's dense buffers IN PLACE: the first prop-down takes H_dev (pitch H) as its
        // operand, the last sweep's launches store straight into V_dev / H_dev - no copy kernels (round 3 moved the
        // states through the pitched workspaces with three copy2d launches per call: 5 % of the sweep benchmark).
        ensure_wt(h);
        chain_begin(h);
        for (int t = 0; t < n_steps; ++t) {
            const bool first = t == 0, last = t == n_steps - 1;
            launch_down(h, first ? H_dev : h->hs.p, first ? h->H : h->hs.ld, B, nullptr, last ? V_dev : h->vs.p,
                        last ? h->V : h->vs.ld, h->cfg.sample_v_states, SITE_V, t);
            launch_up(h, last ? V_dev : h->vs.p, last ? h->V : h->vs.ld, B, nullptr, last ? H_dev : h->hs.p,
                      last ? h->H : h->hs.ld, h->cfg.sample_h_states, SITE_H, t);
        }
        BM_TRY(chain_end(h));
        h->call++;
        BM_HIP(hipGetLastError());
        return 0;
    }
    // fast-binary / Multinomial sweeps: dense user buffers <-> pitched workspaces (hs / vs, which carry the bf16 shadows)
    hipLaunchKernelGGL(copy2d_kernel, dim3(256), dim3(256), 0, h->stream, (const float *)H_dev, h->H, h->hs.p, h->hs.ld, B, h->H);
    if (fast) {
        // fast-binary sweep: both layers are sampled, so every contraction has a {0,1} operand (the caller's hidden
        // states must be a bitmap as well: checked on the device, reported by bm_rbm_sync)
        if (h->W3.rows != h->V) {
            BM_TRY(h->W3.alloc(3, h->V, h->H)); BM_TRY(h->W3t.alloc(3, h->H, h->V));
            BM_TRY(h->hs16.alloc(1, h->maxB, h->H)); BM_TRY(h->vs16.alloc(1, h->maxB, h->V));
            BM_HIP(hipMalloc((void **)&h->nonbinary, sizeof(int)));
            BM_HIP(hipMemsetAsync(h->nonbinary, 0, sizeof(int), h->stream));
        }
        hipLaunchKernelGGL(split3_kernel, dim3(512), dim3(256), 0, h->stream, (const float *)h->W.p, h->W.ld, h->V, h->H,
                           h->W3.p, h->W3.plane_stride(), h->W3.ld, 0);
        hipLaunchKernelGGL(split3_kernel, dim3(512), dim3(256), 0, h->stream, (const float *)h->W.p, h->W.ld, h->V, h->H,
                           h->W3t.p, h->W3t.plane_stride(), h->W3t.ld, 1);
        hipLaunchKernelGGL(shadow16_check_kernel, dim3(256), dim3(256), 0, h->stream, (const float *)h->hs.p, h->hs.ld, B, h->H,
                           h->hs16.p, h->hs16.ld, h->nonbinary);
        h->fast_now = true;
    }
    for (int t = 0; t < n_steps; ++t) {
        launch_down(h, h->hs.p, h->hs.ld, B, nullptr, h->vs.p, h->vs.ld, h->cfg.sample_v_states, SITE_V, t);
        launch_up(h, h->vs.p, h->vs.ld, B, nullptr, h->hs.p, h->hs.ld, h->cfg.sample_h_states, SITE_H, t);
    }
    hipLaunchKernelGGL(copy2d_kernel, dim3(256), dim3(256), 0, h->stream, (const float *)h->hs.p, h->hs.ld, H_dev, h->H, B, h->H);
    hipLaunchKernelGGL(copy2d_kernel, dim3(256), dim3(256), 0, h->stream, (const float *)h->vs.p, h->vs.ld, V_dev, h->V, B, h->V);
    h->call++;
    BM_HIP(hipGetLastError());
    return 0;
}

// the block -> tile map of a launch with tiles_i x tiles_j tiles (host evaluation of the device function, for tests and
// tools): out_ti / out_tj [tiles_i * tiles_j] receive the tile of every block, out_map5 = {xi, xj, gj, tiles_i, tiles_j}
int bm_debug_tile_map(int32_t tiles_i, int32_t tiles_j, double bytes_i, double bytes_j, int32_t *out_ti, int32_t *out_tj,
                      int32_t *out_map5) {
    BM_CHECK(tiles_i >= 1 && tiles_j >= 1 && out_ti && out_tj, "bad arguments");
    const TileMap m = make_tile_map(tiles_i, tiles_j, bytes_i, bytes_j);
    const int nb = tiles_i * tiles_j;
    for (int b = 0; b < nb; ++b) { int ti = -1, tj = -1; tile_of_block(m, b, nb, ti, tj); out_ti[b] = ti; out_tj[b] = tj; }
    if (out_map5) { out_map5[0] = m.xi; out_map5[1] = m.xj; out_map5[2] = m.gj; out_map5[3] = m.tiles_i; out_map5[4] = m.tiles_j; }
    return 0;
}

int bm_rbm_stream(bm_rbm *h, void **out_stream) { *out_stream = (void *)h->stream; return 0; }

int bm_rbm_profile(bm_rbm *h, int32_t enable) {
    BM_HIP(hipStreamSynchronize(h->stream));
    for (auto &r : h->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    h->recs.clear();
    h->prof = enable != 0;
    return 0;
}

int bm_rbm_kernel_times(bm_rbm *h, float *ms6, int32_t *n6) {
    BM_HIP(hipStreamSynchronize(h->stream));
    for (int c = 0; c < BM_NUM_KERNEL_CLASSES; ++c) { ms6[c] = 0.f; n6[c] = 0; }
    for (auto &r : h->recs) {
        float ms = 0.f;
        BM_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        ms6[r.cls] += ms;
        n6[r.cls] += 1;
    }
    return 0;
}

int bm_rbm_chain_stats(bm_rbm *h, int64_t *out3) {
    BM_CHECK(h && out3, "null argument");
    out3[0] = (int64_t)h->chain.launches; out3[1] = (int64_t)h->chain.tiles_expected; out3[2] = (int64_t)chain_mode(h->chain);
    return 0;
}

int bm_rbm_timer_start(bm_rbm *h) { BM_HIP(hipEventRecord(h->ev0, h->stream)); return 0; }
// the two halves of timer_stop: the mark is enqueued inside a timed region, the read (a host wait) after it
int bm_rbm_timer_mark(bm_rbm *h) { BM_HIP(hipEventRecord(h->ev1, h->stream)); return 0; }
int bm_rbm_timer_elapsed(bm_rbm *h, float *out_ms) {
    BM_HIP(hipEventSynchronize(h->ev1));
    BM_HIP(hipEventElapsedTime(out_ms, h->ev0, h->ev1));
    return 0;
}
int bm_rbm_timer_stop(bm_rbm *h, float *out_ms) {
    BM_HIP(hipEventRecord(h->ev1, h->stream));
    BM_HIP(hipEventSynchronize(h->ev1));
    BM_HIP(hipEventElapsedTime(out_ms, h->ev0, h->ev1));
    return 0;
}

}  // extern "C"
