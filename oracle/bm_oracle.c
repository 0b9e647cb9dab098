/* bm_oracle.c — CPU restatement of the reference's RBM/DBM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (boltzmann_machines_amd/)
 * may include, link or call this file; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, as the checker / CPU baseline.
 *
 * What it restates (reference = yell/boltzmann-machines, TensorFlow-1.3 graph):
 *   - CD-k train op                 boltzmann_machines/rbm/base_rbm.py:415-479
 *   - propagations / Gibbs chain    base_rbm.py:329-413
 *   - layer activations + samplers  boltzmann_machines/layers.py:39-51,73-89
 *   - free energies, Gaussian input rbm/rbm.py:17-22,101-116
 *   - metrics (msre, pll, l2)       base_rbm.py:482-517
 *   - DBM Gibbs sweep / MF / PCD / train op / AIS / ELBO
 *                                   boltzmann_machines/dbm.py:385-759
 *
 * The arithmetic of the reference lives in TensorFlow 1.3 (requirements.txt:11),
 * which is not in /root/reference and cannot run here; its op semantics are
 * restated from SURVEY.md App. B/C.  Pinned against the reference's own golden
 * vector: W-init KAT of rbm/tests/test_rbm.py:64-67 (tests/test_oracle.py).
 * Every other value of the train step is "parity unpinned" by the reference
 * (it holds no fixture for them) and is pinned by this oracle only.
 *
 * "Canonical order": every dot product is the sequential chain
 *     acc = 0; for k ascending: acc = fmaf(a[k], b[k], acc)
 * and every column sum is the sequential fp32 sum over rows.  The HIP kernels
 * reproduce exactly this order (v_mfma_f32_16x16x4_f32 is a k-ordered fma
 * chain), so probabilities, sample bitmaps and parameter updates are
 * BIT-IDENTICAL between this file and the GPU.  Build with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ Philox */
/* Philox4x32-10, TF stream convention. */
static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                          uint32_t k0, uint32_t k1, uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct { uint32_t k0, k1, site, call; } orc_key;

static orc_key make_key(uint64_t seed, uint32_t site, uint32_t call) {
    orc_key k = {(uint32_t)seed, (uint32_t)(seed >> 32), site, call};
    return k;
}

static float u32_to_uniform(uint32_t x) {
    union { uint32_t u; float f; } v;
    v.u = 0x3f800000u | (x & 0x007fffffu);
    return v.f - 1.0f;
}

static void philox_block(orc_key key, uint64_t block, uint32_t w[4]) {
    philox4x32_10((uint32_t)block, (uint32_t)(block >> 32), key.site, key.call, key.k0, key.k1, w);
}

static float uniform_at(orc_key key, uint64_t idx) {
    uint32_t w[4];
    philox_block(key, idx >> 2, w);
    return u32_to_uniform(w[idx & 3]);
}

/* TF BoxMullerFloat */
static float normal_at(orc_key key, uint64_t idx) {
    uint32_t w[4];
    philox_block(key, idx >> 2, w);
    const int pr = (int)((idx & 3) >> 1);
    float u1 = u32_to_uniform(w[2 * pr]);
    if (u1 < 1.0e-7f) u1 = 1.0e-7f;
    const float v1 = 6.2831853071795864769f * u32_to_uniform(w[2 * pr + 1]);
    const float r = sqrtf(-2.0f * logf(u1));
    return ((idx & 1) ? cosf(v1) : sinf(v1)) * r;
}

/* exported for tests: raw words, uniforms and normals of a stream */
void orc_philox_words(uint64_t seed, uint32_t site, uint32_t call, uint64_t block0, uint64_t nblocks,
                      uint32_t *out) {
    orc_key k = make_key(seed, site, call);
    for (uint64_t b = 0; b < nblocks; ++b) philox_block(k, block0 + b, out + 4 * b);
}
void orc_uniform(uint64_t seed, uint32_t site, uint32_t call, uint64_t idx0, uint64_t n, float *out) {
    orc_key k = make_key(seed, site, call);
    for (uint64_t i = 0; i < n; ++i) out[i] = uniform_at(k, idx0 + i);
}
void orc_normal(uint64_t seed, uint32_t site, uint32_t call, uint64_t idx0, uint64_t n, float *out) {
    orc_key k = make_key(seed, site, call);
    for (uint64_t i = 0; i < n; ++i) out[i] = normal_at(k, idx0 + i);
}

/* --------------------------------------------------------------- numerics */
/* tf.nn.sigmoid, specified op-by-op (DESIGN.md "Numerics"): only correctly
 * rounded fp32 ops so the GPU reproduces it bit-for-bit. */
static float exp_neg(float a) {
    const float t = a * -1.44269504088896341f;
    const float n = rintf(t);
    float r = fmaf(n, -0.693145751953125f, -a);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    union { uint32_t u; float f; } v;
    v.f = p;
    v.u += ((uint32_t)(int)n) << 23;
    return v.f;
}

float orc_sigmoid(float x) {
    float a = fabsf(x);
    if (a > 80.0f) a = 80.0f;
    const float e = exp_neg(a);
    const float d = 1.0f + e;
    return (x >= 0.0f) ? (1.0f / d) : (e / d);
}

static double softplus_d(double x) { return fmax(x, 0.0) + log1p(exp(-fabs(x))); }

/* ------------------------------------------------------------ contractions */
/* out[j][i] = sum_k Q[j][k] * Pk[k][i]   (k ascending fmaf chain), optionally
 * continuing from a second segment.  Pk is k-major ([K][I], i contiguous). */
static void chain_kmajor(float *acc, const float *Qrow, const float *Pk, int K, int I) {
    for (int k = 0; k < K; ++k) {
        const float q = Qrow[k];
        const float *p = Pk + (size_t)k * I;
        for (int i = 0; i < I; ++i) acc[i] = fmaf(p[i], q, acc[i]);
    }
}

static float *transpose(const float *A, int R, int C) {   /* A[R][C] -> T[C][R] */
    float *T = (float *)malloc((size_t)R * C * sizeof(float));
    for (int r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) T[(size_t)c * R + r] = A[(size_t)r * C + c];
    return T;
}

enum { UNIT_BERNOULLI = 0, UNIT_GAUSSIAN = 1 };

/* One fused "activation" stage = what act_kernel does on the GPU:
 *   z[j][i] = chain(seg1) then chain(seg2);  x = mult*z; b = mult*bias[i]
 *   Bernoulli: m = sigmoid(x + b)            (layers.py:47-48)
 *   Gaussian : m = x*sigma[i] + b            (layers.py:84-86)
 *   states   = sample ? draw(m) : m          (layers.py:34-36,50-51,88-89)
 * P1k [K1][I] and P2k [K2][I] are k-major. */
void orc_act(const float *Q1, int K1, const float *P1k,
             const float *Q2, int K2, const float *P2k,
             int I, int J, const float *bias, const float *sigma, float mult, int kind, int sample,
             float *means, float *states,
             uint64_t seed, uint32_t site, uint32_t call, int64_t row0) {
    const orc_key key = make_key(seed, site, call);
#pragma omp parallel
    {
        float *acc = (float *)malloc((size_t)I * sizeof(float));
#pragma omp for schedule(static)
        for (int j = 0; j < J; ++j) {
            for (int i = 0; i < I; ++i) acc[i] = 0.0f;
            chain_kmajor(acc, Q1 + (size_t)j * K1, P1k, K1, I);
            if (K2 > 0) chain_kmajor(acc, Q2 + (size_t)j * K2, P2k, K2, I);
            for (int i = 0; i < I; ++i) {
                const float x = mult * acc[i];
                const float b = mult * bias[i];
                const float m = (kind == UNIT_BERNOULLI) ? orc_sigmoid(x + b) : (x * sigma[i] + b);
                float s = m;
                if (sample) {
                    const uint64_t idx = (uint64_t)(row0 + j) * (uint64_t)I + (uint64_t)i;
                    if (kind == UNIT_BERNOULLI) s = (uniform_at(key, idx) < m) ? 1.0f : 0.0f;
                    else s = normal_at(key, idx) * sigma[i] + m;
                }
                if (means) means[(size_t)j * I + i] = m;
                if (states) states[(size_t)j * I + i] = s;
            }
        }
        free(acc);
    }
}

/* out[j][i] (+)= sgn * sum_b Qb[b][j] * Pb[b][i]  — outer-product accumulation over rows b,
 * continuing the chain already in `out` when `accumulate` is set */
static void outer_chain(float *out, const float *Qb, int J, const float *Pb, int I, int B, float sgn,
                        int accumulate) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < J; ++j) {
        float *acc = out + (size_t)j * I;
        if (!accumulate) for (int i = 0; i < I; ++i) acc[i] = 0.0f;
        for (int b = 0; b < B; ++b) {
            const float q = sgn * Qb[(size_t)b * J + j];
            const float *p = Pb + (size_t)b * I;
            for (int i = 0; i < I; ++i) acc[i] = fmaf(p[i], q, acc[i]);
        }
    }
}

/* out[c] = sum_b (A[b][c] - Bm[b][c]), sequential over b */
static void colsum_diff(float *out, const float *A, const float *Bm, int B, int C) {
    for (int c = 0; c < C; ++c) out[c] = 0.0f;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            float x = A[(size_t)b * C + c];
            if (Bm) x = x - Bm[(size_t)b * C + c];
            out[c] = out[c] + x;
        }
}

/* ---------------------------------------------------------------- RBM path */
typedef struct {
    int32_t V, H;
    int32_t v_unit, sample_v, sample_h, dbm_first, dbm_last;
    float l2, sp_target, sp_cost, sp_damping, dropout;   /* dropout < 0: off */
} orc_rbm_cfg;

typedef struct { float *W, *vb, *hb, *dW, *dvb, *dhb, *q, *sigma; } orc_rbm_state;

/* chain intermediates, all caller-allocated: Xin [B,V], h0m/h0s/hm/hs [B,H], vm/vs [B,V] */
typedef struct { float *Xin, *h0m, *h0s, *vm, *vs, *hm, *hs; } orc_rbm_work;

enum { SITE_DROPOUT = 1, SITE_H0 = 2, SITE_V = 3, SITE_H = 4, SITE_PLL = 5 };

/* base_rbm.py:417-426 (+ rbm.py:107 for the Gaussian input scaling) */
void orc_rbm_chain(const orc_rbm_cfg *c, const orc_rbm_state *s, const float *X, int B, int k,
                   uint64_t seed, uint32_t call, int64_t row0, orc_rbm_work *w) {
    const int V = c->V, H = c->H;
    const size_t nX = (size_t)B * V;
    for (size_t e = 0; e < nX; ++e) {
        float x = X[e];
        if (c->v_unit == UNIT_GAUSSIAN) x = x / s->sigma[e % (size_t)V];          /* rbm.py:107 */
        w->Xin[e] = x;
    }
    if (c->dropout >= 0.0f) {                                                      /* base_rbm.py:417-418 */
        const orc_key key = make_key(seed, SITE_DROPOUT, call);
        for (size_t e = 0; e < nX; ++e) {
            const float u = uniform_at(key, (uint64_t)row0 * (uint64_t)V + e);
            w->Xin[e] = (w->Xin[e] / c->dropout) * floorf(c->dropout + u);
        }
    }
    const float up = 1.0f + (c->dbm_first ? 1.0f : 0.0f);                          /* :256-260 */
    const float down = 1.0f + (c->dbm_last ? 1.0f : 0.0f);                         /* :261-262 */
    float *Wt = transpose(s->W, V, H);                                             /* Wt[h][v] */
    /* h0 (always sampled, used iff sample_h_states)  :421-423 */
    orc_act(w->Xin, V, s->W, NULL, 0, NULL, H, B, s->hb, NULL, up, UNIT_BERNOULLI, 1,
            w->h0m, w->h0s, seed, SITE_H0, call, row0);
    const float *hstate = c->sample_h ? w->h0s : w->h0m;
    for (int t = 0; t < k; ++t) {                                                  /* :367-378 */
        orc_act(hstate, H, Wt, NULL, 0, NULL, V, B, s->vb, s->sigma, down, c->v_unit, c->sample_v,
                w->vm, w->vs, seed, SITE_V + 16u * (uint32_t)t, call, row0);
        orc_act(w->vs, V, s->W, NULL, 0, NULL, H, B, s->hb, NULL, up, UNIT_BERNOULLI, c->sample_h,
                w->hm, w->hs, seed, SITE_H + 16u * (uint32_t)t, call, row0);
        hstate = w->hs;
    }
    free(Wt);
}

/* raw sums of the gradient estimate: base_rbm.py:447-453,457
 * raw = [ X^T h0 - v^T h_k  (V*H) | sum(X - v) (V) | sum(h0 - h_k) (H) | sum(h_k) (H) ] */
void orc_rbm_raw_grads(const orc_rbm_cfg *c, const orc_rbm_work *w, int B, float *raw) {
    const int V = c->V, H = c->H;
    /* dW_positive - dW_negative (:447-449) as ONE canonical chain: the positive rows
     * X^T h0_means, then the negative rows with the product negated, -(v_states^T h_means) */
    outer_chain(raw, w->Xin, V, w->h0m, H, B, 1.0f, 0);
    outer_chain(raw, w->vs, V, w->hm, H, B, -1.0f, 1);
    float *tail = raw + (size_t)V * H;
    colsum_diff(tail, w->Xin, w->vs, B, V);
    colsum_diff(tail + V, w->h0m, w->hm, B, H);
    colsum_diff(tail + V + H, w->hm, NULL, B, H);
}

/* sparsity + momentum + assign_add: base_rbm.py:455-474.  N = (global) batch rows. */
void orc_rbm_apply(const orc_rbm_cfg *c, orc_rbm_state *s, const float *raw, float N, float lr, float mom) {
    const int V = c->V, H = c->H;
    const float *sv = raw + (size_t)V * H, *sh = sv + V, *sq = sh + H;
    float *pen = (float *)malloc((size_t)H * sizeof(float));
    for (int v = 0; v < V; ++v) {
        const float g = sv[v] / N;
        const float d = lr * (mom * s->dvb[v] + g);
        s->dvb[v] = d;
        s->vb[v] = s->vb[v] + d;
    }
    for (int h = 0; h < H; ++h) {
        const float qn = c->sp_damping * s->q[h] + (1.0f - c->sp_damping) * sq[h];
        s->q[h] = qn;
        pen[h] = c->sp_cost * (qn - c->sp_target);
        float g = sh[h] / N;
        g = g - pen[h];
        const float d = lr * (mom * s->dhb[h] + g);
        s->dhb[h] = d;
        s->hb[h] = s->hb[h] + d;
    }
    for (int v = 0; v < V; ++v)
        for (int h = 0; h < H; ++h) {
            const size_t e = (size_t)v * H + h;
            float g = raw[e] / N;
            g = g - c->l2 * s->W[e];
            g = g - pen[h];
            const float d = lr * (mom * s->dW[e] + g);
            s->dW[e] = d;
            s->W[e] = s->W[e] + d;
        }
    free(pen);
}

/* session.run(train_op) — base_rbm.py:566 */
void orc_rbm_train_step(const orc_rbm_cfg *c, orc_rbm_state *s, const float *X, int B, float lr, float mom,
                        int k, uint64_t seed, uint32_t call, int64_t row0, orc_rbm_work *w) {
    orc_rbm_chain(c, s, X, B, k, seed, call, row0, w);
    float *raw = (float *)malloc(((size_t)c->V * c->H + c->V + 2 * (size_t)c->H) * sizeof(float));
    orc_rbm_raw_grads(c, w, B, raw);
    orc_rbm_apply(c, s, raw, (float)B, lr, mom);
    free(raw);
}

/* batch-mean free energy (double accumulation; tolerance-checked):
 * Bernoulli rbm.py:17-22, Gaussian rbm.py:109-116.  Xin = input AFTER /sigma. */
double orc_rbm_free_energy(const orc_rbm_cfg *c, const orc_rbm_state *s, const float *Xin, int B,
                           const int32_t *flip) {
    const int V = c->V, H = c->H;
    double total = 0.0;
    for (int b = 0; b < B; ++b) {
        const float *x = Xin + (size_t)b * V;
        double t = 0.0;
        for (int v = 0; v < V; ++v) {
            double xv = x[v];
            if (flip && flip[b] == v) xv = 1.0 - xv;                       /* base_rbm.py:503-509 */
            if (c->v_unit == UNIT_GAUSSIAN) {
                const double mu = (double)s->vb[v] / (double)s->sigma[v];
                t += 0.5 * (xv - mu) * (xv - mu);
            } else {
                t -= xv * (double)s->vb[v];
            }
        }
        for (int h = 0; h < H; ++h) {
            double z = s->hb[h];
            for (int v = 0; v < V; ++v) {
                double xv = x[v];
                if (flip && flip[b] == v) xv = 1.0 - xv;
                z += xv * (double)s->W[(size_t)v * H + h];
            }
            t -= softplus_d(z);
        }
        total += t;
    }
    return total / B;
}

/* metrics of base_rbm.py:482-517 from a finished chain: out = [msre, pll, l2_loss, free_energy] */
void orc_rbm_metrics(const orc_rbm_cfg *c, const orc_rbm_state *s, const orc_rbm_work *w, int B,
                     uint64_t seed, uint32_t call, int64_t row0, float *out4, int32_t *flip_out) {
    const int V = c->V, H = c->H;
    double se = 0.0;
    for (size_t e = 0; e < (size_t)B * V; ++e) {
        const double d = (double)w->Xin[e] - (double)w->vm[e];
        se += d * d;
    }
    out4[0] = (float)(se / ((double)B * V));
    double l2 = 0.0;
    for (size_t e = 0; e < (size_t)V * H; ++e) l2 += (double)s->W[e] * (double)s->W[e];
    out4[2] = c->l2 * (float)(0.5 * l2);
    int32_t *flip = (int32_t *)malloc((size_t)B * sizeof(int32_t));
    const orc_key key = make_key(seed, SITE_PLL, call);
    for (int b = 0; b < B; ++b) {                 /* tf.random_uniform int32: minval + u32 % range */
        const uint64_t idx = (uint64_t)row0 + (uint64_t)b;
        uint32_t wd[4];
        philox_block(key, idx >> 2, wd);
        flip[b] = (int32_t)(wd[idx & 3] % (uint32_t)V);
        if (flip_out) flip_out[b] = flip[b];
    }
    const double fe = orc_rbm_free_energy(c, s, w->Xin, B, NULL);
    const double fe2 = orc_rbm_free_energy(c, s, w->Xin, B, flip);
    const double d = fe2 - fe;
    out4[1] = (float)((double)V * -softplus_d(-d));       /* V * log_sigmoid(F(x~) - F(x))  :511-512 */
    out4[3] = (float)fe;
    free(flip);
}
