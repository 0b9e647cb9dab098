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
 *   - MultinomialRBM hidden layer + free energy
 *                                   rbm/rbm.py:25-65, layers.py:54-70
 *   - the float64 RBM path (dtype of base/mixin.py:15; rbm/tests/test_rbm.py:53-56,70-73)
 *                                   end of this file
 *
 * The arithmetic of the reference lives in TensorFlow 1.3 (requirements.txt:11),
 * which is not in /root/reference and cannot run here; its op semantics are
 * restated from SURVEY.md App. B/C.  Pinned against the reference's own golden
 * vector: W-init KAT of rbm/tests/test_rbm.py:64-67 (tests/test_oracle.py).
 * Every other value of the train step is "parity unpinned" by the reference
 * (it holds no fixture for them) and is pinned by this oracle only.
 *
 * "Canonical order": every dot product of a matmul is ONE sequential fma chain
 *     acc = 0; for k in ORDER: acc = fmaf(a[k], b[k], acc)
 * whose ORDER visits every aligned block of 16 k as  k = 16m + 4g + j  for j = 0..3 (outer), g = 0..3 (inner)
 * - i.e. 0,4,8,12, 1,5,9,13, 2,6,10,14, 3,7,11,15, then the next block (k >= K skipped; the second segment of
 * a two-segment contraction starts its own blocks).  The order of a float32 matmul's accumulation is an
 * implementation detail of the reference's backend (Eigen / cuBLAS inside TensorFlow 1.3: blocked, unknowable
 * and certainly not sequential either); the engine and this file fix it, identically, to the order in which
 * v_mfma_f32_16x16x4_f32 consumes 16-byte chunks of k-contiguous rows (csrc/bm_gemm.h), so that
 * probabilities, sample bitmaps and parameter updates are BIT-IDENTICAL between this file and the GPU.
 * tests/np_reference.py (float64, NumPy's own order) checks that nothing depends on the choice beyond
 * float32 round-off.  Every column sum is the sequential fp32 sum over rows.  Build with -ffp-contract=off.
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
/* ln(u), u in [2^-24, 1), and (sin, cos)(2 pi u), u in [0, 1): the operation-by-operation definitions of
 * csrc/bm_rng.h (`pin_log_unit`, `pin_sincos_2pi`) - correctly rounded fp32 operations only, so that every Normal
 * draw of the engine is reproduced bit for bit.  TF's own BoxMullerFloat calls the platform's logf / sinf / cosf:
 * unknowable in the last bit, this definition is within 1.2e-7 of them. */
static float pin_log_unit(float u) {
    union { uint32_t u; float f; } v;
    v.f = u;
    int e = (int)(v.u >> 23) - 127;
    v.u = (v.u & 0x007fffffu) | 0x3f800000u;
    float m = v.f;
    if (m > 1.41421356237309504880f) { m = m * 0.5f; e = e + 1; }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float t = s * s;
    float p = 1.0f / 11.0f;
    p = fmaf(p, t, 1.0f / 9.0f);
    p = fmaf(p, t, 1.0f / 7.0f);
    p = fmaf(p, t, 1.0f / 5.0f);
    p = fmaf(p, t, 1.0f / 3.0f);
    p = fmaf(p, t, 1.0f);
    const float lm = (2.0f * s) * p;
    const float ef = (float)e;
    return fmaf(ef, 0.693145751953125f, fmaf(ef, 1.42860682030941723212e-6f, lm));
}
static void pin_sincos_2pi(float u, float *sn, float *cs) {
    const float t = u * 4.0f;
    int q = (int)t;
    float f = t - (float)q;
    if (f > 0.5f) { f = f - 1.0f; q = q + 1; }
    const float x = f * 1.57079632679489661923f;
    const float x2 = x * x;
    float ps = 2.75573192239858906526e-6f;
    ps = fmaf(ps, x2, -1.98412698412698412698e-4f);
    ps = fmaf(ps, x2, 8.33333333333333333333e-3f);
    ps = fmaf(ps, x2, -1.66666666666666666667e-1f);
    const float sx = fmaf(x * x2, ps, x);
    float pc = -2.75573192239858906526e-7f;
    pc = fmaf(pc, x2, 2.48015873015873015873e-5f);
    pc = fmaf(pc, x2, -1.38888888888888888889e-3f);
    pc = fmaf(pc, x2, 4.16666666666666666667e-2f);
    pc = fmaf(pc, x2, -0.5f);
    const float cx = fmaf(pc, x2, 1.0f);
    switch (q & 3) {
        case 0: *sn = sx; *cs = cx; break;
        case 1: *sn = cx; *cs = -sx; break;
        case 2: *sn = -sx; *cs = -cx; break;
        default: *sn = -cx; *cs = sx; break;
    }
}
float orc_pin_log_unit(float u) { return pin_log_unit(u); }
void orc_pin_sincos_2pi(float u, float *sn, float *cs) { pin_sincos_2pi(u, sn, cs); }

static float normal_at(orc_key key, uint64_t idx) {
    uint32_t w[4];
    philox_block(key, idx >> 2, w);
    const int pr = (int)((idx & 3) >> 1);
    float u1 = u32_to_uniform(w[2 * pr]);
    if (u1 < 1.0e-7f) u1 = 1.0e-7f;
    const float r = sqrtf(-2.0f * pin_log_unit(u1));
    float sn, cs;
    pin_sincos_2pi(u32_to_uniform(w[2 * pr + 1]), &sn, &cs);
    return ((idx & 1) ? cs : sn) * r;
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

/* "Reference arithmetic" (orc_dbm_cfg.sigmoid_literal; engine: bm_dbm_set_sigmoid_literal): tf.nn.sigmoid as TensorFlow 1.3
 * evaluates it on the CPU (layers.py:47-48): float32 1 / (1 + exp(-x)) (Eigen scalar_sigmoid_op), exp = Eigen 3.3's
 * pexp<Packet4f> - the Cephes expf scheme, every pmadd a rounded multiply + a rounded add (SSE build, no FMA).  TensorFlow
 * (requirements.txt:11 `tensorflow-gpu~=1.3.0`) and its Eigen are not in /root/reference: restated from the published
 * algorithm, operation by operation, identically in csrc/bm_numerics.h (exp_eigen) and tests/tf1_shim (tf.sigmoid).
 * Built with -ffp-contract=off (Makefile): nothing below may fuse. */
float orc_exp_eigen(float x0) {
    float x = fminf(x0, 88.3762626647950f);
    x = fmaxf(x, -88.3762626647949f);
    float fx = x * 1.44269504088896341f;
    fx = fx + 0.5f;
    fx = floorf(fx);
    const float tmp = fx * 0.693359375f;
    const float zz = fx * -2.12194440e-4f;
    x = x - tmp;
    x = x - zz;
    const float z = x * x;
    float y = 1.9875691500E-4f;
    y = y * x; y = y + 1.3981999507E-3f;
    y = y * x; y = y + 8.3334519073E-3f;
    y = y * x; y = y + 4.1665795894E-2f;
    y = y * x; y = y + 1.6666665459E-1f;
    y = y * x; y = y + 5.0000001201E-1f;
    y = y * z; y = y + x;
    y = y + 1.0f;
    union { uint32_t u; float f; } p2;
    p2.u = (uint32_t)((int32_t)fx + 127) << 23;
    return fmaxf(y * p2.f, x0);
}
float orc_sigmoid_literal(float x) { return 1.0f / (1.0f + orc_exp_eigen(-x)); }

static double softplus_d(double x) { return fmax(x, 0.0) + log1p(exp(-fabs(x))); }

/* ------------------------------------------------------------ contractions */
/* out[j][i] = sum_k Q[j][k] * Pk[k][i]   (k ascending fmaf chain), optionally
 * continuing from a second segment.  Pk is k-major ([K][I], i contiguous). */
static void chain_kmajor(float *acc, const float *Qrow, const float *Pk, int K, int I) {
    for (int kb = 0; kb < K; kb += 16)                      /* canonical order (file header) */
        for (int j = 0; j < 4; ++j)
            for (int g = 0; g < 4; ++g) {
                const int k = kb + 4 * g + j;
                if (k >= K) continue;
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

enum { UNIT_BERNOULLI = 0, UNIT_GAUSSIAN = 1, UNIT_MULTINOMIAL = 2,
       UNIT_BERNOULLI_LIT = 4 /* orc_act2 only: Bernoulli with orc_sigmoid_literal */ };

/* One fused "activation" stage = what act_kernel does on the GPU:
 *   z[j][i] = chain(seg1) then chain(seg2);  x = mult*z; b = mult*bias[i]
 *   Bernoulli: m = sigmoid(x + b)            (layers.py:47-48)
 *   Gaussian : m = x*sigma[i] + b            (layers.py:84-86)
 *   states   = sample ? draw(m) : m          (layers.py:34-36,50-51,88-89)
 * P1k [K1][I] and P2k [K2][I] are k-major. */
void orc_act2(const float *Q1, int K1, const float *P1k,
              const float *Q2, int K2, const float *P2k,
              int I, int J, const float *bias, const float *sigma, float mult, float bmult, int kind, int sample,
              float *means, float *states,
              uint64_t seed, uint32_t site, uint32_t call, int64_t row0);

void orc_act(const float *Q1, int K1, const float *P1k,
             const float *Q2, int K2, const float *P2k,
             int I, int J, const float *bias, const float *sigma, float mult, int kind, int sample,
             float *means, float *states,
             uint64_t seed, uint32_t site, uint32_t call, int64_t row0) {
    orc_act2(Q1, K1, P1k, Q2, K2, P2k, I, J, bias, sigma, mult, mult, kind, sample, means, states, seed, site, call, row0);
}

/* MultinomialLayer (layers.py:54-70) on one row of logits l[0..I):
 *   means = M * softmax(l)                                   activation, :64-65
 *   states = counts of M categorical draws with p = softmax   _sample,   :67-69
 *            (Multinomial(total_count=M, probs=means/reduce_sum(means)): tf.multinomial
 *             renormalises, so the whole-batch reduce_sum of :68 cancels)
 * Specified like the sigmoid, operation by operation, so the GPU reproduces it bit for bit:
 *   mx = max_i l[i];  e[i] = exp_neg(min(mx - l[i], 80));  c[i] = c[i-1] + e[i]  (sequential fp32)
 *   S = c[I-1];  means[i] = M * (e[i] / S)
 *   draw d (0 <= d < M): u = uniform(flat index row*M + d);  t = u * S;
 *                        category = smallest i with c[i] > t   (exists: t < S)
 * TF's own multinomial kernel consumes its RNG stream in an undocumented order: parity unpinned,
 * pinned by this definition. */
static void softmax_multinomial_row(const float *l, int I, int M, int sample, float *means, float *states,
                                    orc_key key, uint64_t row, float *e, float *c) {
    float mx = l[0];
    for (int i = 1; i < I; ++i) mx = fmaxf(mx, l[i]);
    float run = 0.0f;
    for (int i = 0; i < I; ++i) {
        float a = mx - l[i];
        if (a > 80.0f) a = 80.0f;
        e[i] = exp_neg(a);
        run = run + e[i];
        c[i] = run;
    }
    const float S = c[I - 1];
    for (int i = 0; i < I; ++i) {
        const float m = (float)M * (e[i] / S);
        if (means) means[i] = m;
        if (states) states[i] = sample ? 0.0f : m;
    }
    if (sample && states) {
        for (int d = 0; d < M; ++d) {
            const float u = uniform_at(key, row * (uint64_t)M + (uint64_t)d);
            const float t = u * S;
            int lo = 0, hi = I - 1;                 /* smallest i with c[i] > t */
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (c[mid] > t) hi = mid; else lo = mid + 1;
            }
            states[lo] = states[lo] + 1.0f;
        }
    }
}

/* bmult: multiplier of the bias (== mult except in the mean-field init, dbm.py:434-446) */
void orc_act2(const float *Q1, int K1, const float *P1k,
              const float *Q2, int K2, const float *P2k,
              int I, int J, const float *bias, const float *sigma, float mult, float bmult, int kind, int sample,
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
            if (kind >= 16) {               /* Multinomial: kind = 16 + n_samples, logits x + b */
                const int M = kind - 16;
                float *e = (float *)malloc(2 * (size_t)I * sizeof(float));
                for (int i = 0; i < I; ++i) acc[i] = mult * acc[i] + bmult * bias[i];
                softmax_multinomial_row(acc, I, M, sample, means ? means + (size_t)j * I : NULL,
                                        states ? states + (size_t)j * I : NULL, key, (uint64_t)(row0 + j), e, e + I);
                free(e);
                continue;
            }
            for (int i = 0; i < I; ++i) {
                const float x = mult * acc[i];
                const float b = bmult * bias[i];
                const int bern = (kind == UNIT_BERNOULLI || kind == UNIT_BERNOULLI_LIT);
                const float m = (kind == UNIT_BERNOULLI) ? orc_sigmoid(x + b)
                              : (kind == UNIT_BERNOULLI_LIT) ? orc_sigmoid_literal(x + b) : (x * sigma[i] + b);
                float s = m;
                if (sample) {
                    const uint64_t idx = (uint64_t)(row0 + j) * (uint64_t)I + (uint64_t)i;
                    if (bern) s = (uniform_at(key, idx) < m) ? 1.0f : 0.0f;
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
        for (int bb = 0; bb < B; bb += 16)                  /* canonical order over the rows b (file header) */
            for (int jj = 0; jj < 4; ++jj)
                for (int g = 0; g < 4; ++g) {
                    const int b = bb + 4 * g + jj;
                    if (b >= B) continue;
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
    int32_t h_unit, n_samples;                           /* UNIT_MULTINOMIAL: MultinomialRBM (rbm.py:25-65) */
} orc_rbm_cfg;

typedef struct { float *W, *vb, *hb, *dW, *dvb, *dhb, *q, *sigma; } orc_rbm_state;

/* chain intermediates, all caller-allocated: Xin [B,V], h0m/h0s/hm/hs [B,H], vm/vs [B,V] */
typedef struct { float *Xin, *h0m, *h0s, *vm, *vs, *hm, *hs; } orc_rbm_work;

enum { SITE_DROPOUT = 1, SITE_H0 = 2, SITE_V = 3, SITE_H = 4, SITE_PLL = 5, SITE_FE = 6 };

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
    const int hkind = (c->h_unit == UNIT_MULTINOMIAL) ? 16 + c->n_samples : UNIT_BERNOULLI;
    orc_act(w->Xin, V, s->W, NULL, 0, NULL, H, B, s->hb, NULL, up, hkind, 1,
            w->h0m, w->h0s, seed, SITE_H0, call, row0);
    const float *hstate = c->sample_h ? w->h0s : w->h0m;
    for (int t = 0; t < k; ++t) {                                                  /* :367-378 */
        orc_act(hstate, H, Wt, NULL, 0, NULL, V, B, s->vb, s->sigma, down, c->v_unit, c->sample_v,
                w->vm, w->vs, seed, SITE_V + 16u * (uint32_t)t, call, row0);
        orc_act(w->vs, V, s->W, NULL, 0, NULL, H, B, s->hb, NULL, up, hkind, c->sample_h,
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

/* h_hat ~ Multinomial(total_count = M, logits = ones[K]).sample()  (rbm.py:58): counts of M uniform
 * category draws, category = floor(u * K).  Stream: site SITE_FE + 16 t. */
static void multinomial_uniform_counts(int K, int M, uint64_t seed, uint32_t call, uint32_t t, double *hhat) {
    const orc_key key = make_key(seed, SITE_FE + 16u * t, call);
    for (int k = 0; k < K; ++k) hhat[k] = 0.0;
    for (int d = 0; d < M; ++d) {
        int idx = (int)(uniform_at(key, (uint64_t)d) * (float)K);
        if (idx > K - 1) idx = K - 1;
        hhat[idx] += 1.0;
    }
}

/* batch-mean free energy (double accumulation; tolerance-checked):
 * Bernoulli rbm.py:17-22, Gaussian rbm.py:109-116, Multinomial rbm.py:52-62 (a fresh random
 * h_hat per evaluation; `t` selects its stream).  Xin = input AFTER /sigma. */
double orc_rbm_free_energy_ex(const orc_rbm_cfg *c, const orc_rbm_state *s, const float *Xin, int B,
                              const int32_t *flip, uint64_t seed, uint32_t call, uint32_t t) {
    const int V = c->V, H = c->H;
    double total = 0.0;
    double *hhat = NULL;
    if (c->h_unit == UNIT_MULTINOMIAL) {
        hhat = (double *)malloc((size_t)H * sizeof(double));
        multinomial_uniform_counts(H, c->n_samples, seed, call, t, hhat);
    }
    for (int b = 0; b < B; ++b) {
        const float *x = Xin + (size_t)b * V;
        double tt = 0.0;
        for (int v = 0; v < V; ++v) {
            double xv = x[v];
            if (flip && flip[b] == v) xv = 1.0 - xv;                       /* base_rbm.py:503-509 */
            if (c->v_unit == UNIT_GAUSSIAN) {
                const double mu = (double)s->vb[v] / (double)s->sigma[v];
                tt += 0.5 * (xv - mu) * (xv - mu);
            } else {
                tt -= xv * (double)s->vb[v];
            }
        }
        for (int h = 0; h < H; ++h) {
            double z = hhat ? 0.0 : s->hb[h];
            for (int v = 0; v < V; ++v) {
                double xv = x[v];
                if (flip && flip[b] == v) xv = 1.0 - xv;
                z += xv * (double)s->W[(size_t)v * H + h];
            }
            tt -= hhat ? z * hhat[h] : softplus_d(z);                     /* rbm.py:57-60: T3 = -(vW).h_hat */
        }
        total += tt;
    }
    double fe = total / B;
    if (hhat) {
        const double M = c->n_samples, K = H;
        fe += -lgamma(M + K) + lgamma(M + 1.0) + lgamma(K);                /* rbm.py:61 */
        free(hhat);
    }
    return fe;
}

double orc_rbm_free_energy(const orc_rbm_cfg *c, const orc_rbm_state *s, const float *Xin, int B,
                           const int32_t *flip) {
    return orc_rbm_free_energy_ex(c, s, Xin, B, flip, 0, 0, 0);
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
    /* Bernoulli/Gaussian: the three evaluations of :511-516 coincide pairwise; Multinomial draws a
     * fresh h_hat in each _free_energy() call (streams t = 0: free_energy_op, 1: F(x), 2: F(x~)) */
    const int mn = c->h_unit == UNIT_MULTINOMIAL;
    const double fe = orc_rbm_free_energy_ex(c, s, w->Xin, B, NULL, seed, call, 0);
    const double fe1 = mn ? orc_rbm_free_energy_ex(c, s, w->Xin, B, NULL, seed, call, 1) : fe;
    const double fe2 = orc_rbm_free_energy_ex(c, s, w->Xin, B, flip, seed, call, 2);
    const double d = fe2 - fe1;
    out4[1] = (float)((double)V * -softplus_d(-d));       /* V * log_sigmoid(F(x~) - F(x))  :511-512 */
    out4[3] = (float)fe;
    free(flip);
}

/* ================================================================= DBM path */
#define ORC_MAXL 4
typedef struct {
    int32_t L, V, n[ORC_MAXL];            /* hidden layer sizes */
    int32_t v_unit, sample_v, sample_h[ORC_MAXL];
    int32_t N, M, max_mf;
    float mf_tol, l2, max_norm, sp_target[ORC_MAXL], sp_cost[ORC_MAXL], sp_damping;
    int32_t h_unit[ORC_MAXL], n_samples[ORC_MAXL];   /* hidden layer kinds (layers.py:39-70): Bernoulli | Multinomial(n_samples) */
    int32_t sigmoid_literal;              /* 1: Bernoulli layers use orc_sigmoid_literal (the reference's float32 tf.sigmoid) */
} orc_dbm_cfg;

/* activation kind of hidden layer i for orc_act2 (Multinomial: 16 + n_samples) */
static int hkind(const orc_dbm_cfg *c, int i) {
    return (c->h_unit[i] == UNIT_MULTINOMIAL) ? 16 + c->n_samples[i] : (c->sigmoid_literal ? UNIT_BERNOULLI_LIT : UNIT_BERNOULLI);
}
/* the same for the visible layer / an all-Bernoulli stack (AIS) */
static int vkind(const orc_dbm_cfg *c) {
    return (c->v_unit == UNIT_BERNOULLI && c->sigmoid_literal) ? UNIT_BERNOULLI_LIT : c->v_unit;
}
static int bkind(const orc_dbm_cfg *c) { return c->sigmoid_literal ? UNIT_BERNOULLI_LIT : UNIT_BERNOULLI; }

typedef struct {
    float *W[ORC_MAXL], *dW[ORC_MAXL], *hb[ORC_MAXL], *dhb[ORC_MAXL], *q[ORC_MAXL], *mm[ORC_MAXL];
    float *mu[ORC_MAXL], *mu_new[ORC_MAXL], *H[ORC_MAXL], *H_new[ORC_MAXL];
    float *vb, *dvb, *sigma, *v, *v_new;
    float *wnorm[ORC_MAXL];
} orc_dbm_state;

enum { SITE_DBM_H = 8, SITE_DBM_V = 12, SITE_AIS_X0 = 13 };

static int dn(const orc_dbm_cfg *c, int i) { return i == 0 ? c->V : c->n[i - 1]; }   /* n[0]=V, n[i+1]=hidden i */

/* `_make_gibbs_step` (dbm.py:385-427): bottom-up sweep with NEW below / OLD above */
static void dbm_sweep(const orc_dbm_cfg *c, const orc_dbm_state *s, int J, const float *vin, float *const *Hin,
                      float *vout, float *const *Hout, int update_v, int sample, int t,
                      uint64_t seed, uint32_t call, int64_t row0) {
    const int L = c->L;
    for (int i = 0; i < L; ++i) {
        const float *below = (i == 0) ? vin : Hout[i - 1];
        const int Kb = dn(c, i), I = dn(c, i + 1);
        float *Wt = NULL; const float *above = NULL; int Ka = 0;
        if (i + 1 < L) { Ka = dn(c, i + 2); Wt = transpose(s->W[i + 1], I, Ka); above = Hin[i + 1]; }
        const int smp = sample && c->sample_h[i];
        orc_act2(below, Kb, s->W[i], above, Ka, Wt, I, J, s->hb[i], NULL, 1.0f, 1.0f, hkind(c, i), smp,
                 NULL, Hout[i], seed, SITE_DBM_H + (uint32_t)i + 16u * (uint32_t)t, call, row0);
        free(Wt);
    }
    if (update_v) {
        float *Wt0 = transpose(s->W[0], c->V, dn(c, 1));
        const int smp = sample && c->sample_v;
        orc_act2(Hout[0], dn(c, 1), Wt0, NULL, 0, NULL, c->V, J, s->vb, s->sigma, 1.0f, 1.0f, vkind(c), smp,
                 NULL, vout, seed, SITE_DBM_V + 16u * (uint32_t)t, call, row0);
        free(Wt0);
    }
}

static float max_abs_diff(const float *a, const float *b, size_t n) {
    float m = 0.0f;
    for (size_t e = 0; e < n; ++e) { const float d = fabsf(a[e] - b[e]); if (d > m) m = d; }
    return m;
}

/* `_make_mf` (dbm.py:429-478); result in s->mu, init values in s->mu_new; returns the sweeps run */
int orc_dbm_mean_field(const orc_dbm_cfg *c, orc_dbm_state *s, const float *X) {
    const int L = c->L, N = c->N;
    for (int i = 0; i < L; ++i) {                              /* approx-inference init :434-446 */
        const float *below = (i == 0) ? X : s->mu_new[i - 1];
        const float mult = (i == 0 || i < L - 1) ? 2.0f : 1.0f;
        orc_act2(below, dn(c, i), s->W[i], NULL, 0, NULL, dn(c, i + 1), N, s->hb[i], NULL, mult, 1.0f,
                 hkind(c, i), 0, s->mu_new[i], NULL, 0, 0, 0, 0);
    }
    float diff = 0.0f;
    for (int i = 0; i < L; ++i) {
        const float d = max_abs_diff(s->mu[i], s->mu_new[i], (size_t)N * dn(c, i + 1));
        if (d > diff) diff = d;
    }
    float *cur[ORC_MAXL], *alt[ORC_MAXL];
    for (int i = 0; i < L; ++i) {
        cur[i] = s->mu[i];
        alt[i] = (float *)malloc((size_t)N * dn(c, i + 1) * sizeof(float));
    }
    int step = 0;
    while (step < c->max_mf && diff > c->mf_tol) {             /* cond :449-452, body :454-457 */
        dbm_sweep(c, s, N, X, cur, NULL, alt, 0, 0, 0, 0, 0, 0);
        diff = 0.0f;
        for (int i = 0; i < L; ++i) {
            const float d = max_abs_diff(cur[i], alt[i], (size_t)N * dn(c, i + 1));
            if (d > diff) diff = d;
        }
        for (int i = 0; i < L; ++i) { float *t = cur[i]; cur[i] = alt[i]; alt[i] = t; }
        ++step;
    }
    for (int i = 0; i < L; ++i) {                              /* mu.assign(result) :477 */
        if (cur[i] != s->mu[i]) { memcpy(s->mu[i], cur[i], (size_t)N * dn(c, i + 1) * sizeof(float)); free(cur[i]); }
        else free(alt[i]);
    }
    return step;
}

/* `_make_particles_update` (dbm.py:480-509): k sweeps with swap; latest state ends in v/H */
void orc_dbm_particles(const orc_dbm_cfg *c, orc_dbm_state *s, int k, int sample,
                       uint64_t seed, uint32_t call, int64_t prow0) {
    for (int t = 0; t < k; ++t) {
        dbm_sweep(c, s, c->M, s->v, s->H, s->v_new, s->H_new, 1, sample, t, seed, call, prow0);
        float *tv = s->v; s->v = s->v_new; s->v_new = tv;
        for (int i = 0; i < c->L; ++i) { float *th = s->H[i]; s->H[i] = s->H_new[i]; s->H_new[i] = th; }
    }
}

/* reconstruction sigma(mu0 W0^T + vb) (dbm.py:625-628) */
void orc_dbm_reconstruct_from_mu(const orc_dbm_cfg *c, const orc_dbm_state *s, float *R) {
    float *Wt0 = transpose(s->W[0], c->V, dn(c, 1));
    orc_act2(s->mu[0], dn(c, 1), Wt0, NULL, 0, NULL, c->V, c->N, s->vb, s->sigma, 1.0f, 1.0f, vkind(c), 0,
             R, NULL, 0, 0, 0, 0);
    free(Wt0);
}

/* gradients + sparsity + momentum + max-norm (dbm.py:550-621) */
static void dbm_apply_update(const orc_dbm_cfg *c, orc_dbm_state *s, const float *X, float lr, float mom) {
    const int L = c->L;
    const float N = (float)c->N, M = (float)c->M;
    float *sx = (float *)malloc(c->V * sizeof(float)), *sv = (float *)malloc(c->V * sizeof(float));
    colsum_diff(sx, X, NULL, c->N, c->V);
    colsum_diff(sv, s->v, NULL, c->M, c->V);
    float *pen[ORC_MAXL];
    /* the W update reads the gradients of the PRE-update state: compute all raw sums first */
    float *pos[ORC_MAXL], *neg[ORC_MAXL];
    for (int i = 0; i < L; ++i) {
        const int J = dn(c, i), I = dn(c, i + 1);
        pos[i] = (float *)malloc((size_t)J * I * sizeof(float));
        neg[i] = (float *)malloc((size_t)J * I * sizeof(float));
        outer_chain(pos[i], (i == 0) ? X : s->mu[i - 1], J, s->mu[i], I, c->N, 1.0f, 0);      /* :556,565 */
        outer_chain(neg[i], (i == 0) ? s->v : s->H[i - 1], J, s->H[i], I, c->M, 1.0f, 0);    /* :557,566 */
    }
    for (int v = 0; v < c->V; ++v) {                                                          /* :553, :597-600 */
        const float g = sx[v] / N - sv[v] / M;
        const float d = lr * (mom * s->dvb[v] + g);
        s->dvb[v] = d;
        s->vb[v] = s->vb[v] + d;
    }
    for (int i = 0; i < L; ++i) {
        const int n = dn(c, i + 1);
        float *smu = (float *)malloc(n * sizeof(float)), *sH = (float *)malloc(n * sizeof(float));
        colsum_diff(smu, s->mu[i], NULL, c->N, n);
        colsum_diff(sH, s->H[i], NULL, c->M, n);
        pen[i] = (float *)malloc(n * sizeof(float));
        for (int h = 0; h < n; ++h) {
            float g = smu[h] / N - sH[h] / M;                                                 /* :573-576 */
            const float qn = c->sp_damping * s->q[i][h] + (1.0f - c->sp_damping) * sH[i];     /* :582-584 scalar index quirk */
            const float mn = c->sp_damping * s->mm[i][h] + (1.0f - c->sp_damping) * smu[i];   /* :585-587 */
            s->q[i][h] = qn;
            s->mm[i][h] = mn;
            const float p1 = c->sp_cost[i] * (qn - c->sp_target[i]);
            const float p2 = c->sp_cost[i] * (mn - c->sp_target[i]);
            pen[i][h] = p1 + p2;
            g = g - pen[i][h];
            const float d = lr * (mom * s->dhb[i][h] + g);
            s->dhb[i][h] = d;
            s->hb[i][h] = s->hb[i][h] + d;
        }
        free(smu); free(sH);
    }
    for (int i = 0; i < L; ++i) {
        const int J = dn(c, i), I = dn(c, i + 1);
        for (int j = 0; j < J; ++j)
            for (int h = 0; h < I; ++h) {
                const size_t e = (size_t)j * I + h;
                float g = pos[i][e] / N - neg[i][e] / M;                                      /* :556-558 */
                g = g - c->l2 * s->W[i][e];
                g = g - pen[i][h];                                                            /* :590 */
                const float d = lr * (mom * s->dW[i][e] + g);                                 /* :604 */
                s->dW[i][e] = d;
                s->W[i][e] = s->W[i][e] + d;                                                  /* :605 */
            }
        for (int h = 0; h < I; ++h) {                                                         /* max-norm :511-513,606-607 */
            float acc = 0.0f;
            for (int j = 0; j < J; ++j) { const float w = s->W[i][(size_t)j * I + h]; acc = fmaf(w, w, acc); }
            const float nrm = sqrtf(acc);
            const float num = fminf(nrm, c->max_norm), den = fmaxf(nrm, 1e-8f);
            for (int j = 0; j < J; ++j) {
                const size_t e = (size_t)j * I + h;
                s->W[i][e] = (s->W[i][e] * num) / den;
            }
            if (s->wnorm[i]) s->wnorm[i][h] = nrm;
        }
        free(pos[i]); free(neg[i]); free(pen[i]);
    }
    free(sx); free(sv);
}

/* session.run(train_op) — dbm.py:805.  Returns executed mean-field sweeps; *msre as dbm.py:625-630 */
int orc_dbm_train_step(const orc_dbm_cfg *c, orc_dbm_state *s, const float *X, float lr, float mom, int k,
                       uint64_t seed, uint32_t call, int64_t prow0, float *msre) {
    const int nmf = orc_dbm_mean_field(c, s, X);
    orc_dbm_particles(c, s, k, 1, seed, call, prow0);
    if (msre) {
        float *R = (float *)malloc((size_t)c->N * c->V * sizeof(float));
        orc_dbm_reconstruct_from_mu(c, s, R);
        double se = 0.0;
        for (size_t e = 0; e < (size_t)c->N * c->V; ++e) { const double d = (double)X[e] - (double)R[e]; se += d * d; }
        *msre = (float)(se / ((double)c->N * c->V));
        free(R);
    }
    dbm_apply_update(c, s, X, lr, mom);
    return nmf;
}

/* sample_v op (dbm.py:641-648): k sampled sweeps (assigned), then k mean sweeps whose v is assigned */
void orc_dbm_sample_v(const orc_dbm_cfg *c, orc_dbm_state *s, int k, uint64_t seed, uint32_t call, int64_t prow0) {
    orc_dbm_particles(c, s, k, 1, seed, call, prow0);
    const int L = c->L, M = c->M;
    float *Hin[ORC_MAXL], *Hout[ORC_MAXL], *vin, *vout;
    vin = (float *)malloc((size_t)M * c->V * sizeof(float)); vout = (float *)malloc((size_t)M * c->V * sizeof(float));
    memcpy(vin, s->v, (size_t)M * c->V * sizeof(float));
    for (int i = 0; i < L; ++i) {
        const size_t n = (size_t)M * dn(c, i + 1);
        Hin[i] = (float *)malloc(n * sizeof(float)); Hout[i] = (float *)malloc(n * sizeof(float));
        memcpy(Hin[i], s->H[i], n * sizeof(float));
    }
    for (int t = 0; t < k; ++t) {
        dbm_sweep(c, s, M, vin, Hin, vout, Hout, 1, 0, k + t, seed, call, prow0);
        float *tv = vin; vin = vout; vout = tv;
        for (int i = 0; i < L; ++i) { float *th = Hin[i]; Hin[i] = Hout[i]; Hout[i] = th; }
    }
    memcpy(s->v, vin, (size_t)M * c->V * sizeof(float));
    free(vin); free(vout);
    for (int i = 0; i < L; ++i) { free(Hin[i]); free(Hout[i]); }
}

/* log p*_beta(x) per chain (dbm.py:650-660), double accumulation */
static void ais_log_p(const orc_dbm_cfg *c, const orc_dbm_state *s, const float *x, int R, float beta, double *out) {
    const int V = c->V, H1 = c->n[0], H2 = c->n[1];
#pragma omp parallel for schedule(static)
    for (int r = 0; r < R; ++r) {
        const float *xr = x + (size_t)r * H1;
        double t1 = 0.0;
        for (int h = 0; h < H1; ++h) t1 += (double)xr[h] * (double)s->hb[0][h];
        double lp = t1 * (double)beta;
        for (int v = 0; v < V; ++v) {
            double z = s->vb[v];
            for (int h = 0; h < H1; ++h) z += (double)xr[h] * (double)s->W[0][(size_t)v * H1 + h];
            lp += softplus_d(z * (double)beta);
        }
        for (int k2 = 0; k2 < H2; ++k2) {
            double z = s->hb[1][k2];
            for (int h = 0; h < H1; ++h) z += (double)xr[h] * (double)s->W[1][(size_t)h * H2 + k2];
            lp += softplus_d(z * (double)beta);
        }
        out[r] = lp;
    }
}

/* AIS (dbm.py:696-736) for the 2-layer Bernoulli DBM; values[r] = log Z estimate of chain r */
void orc_dbm_ais(const orc_dbm_cfg *c, const orc_dbm_state *s, int n_betas, int R, int k,
                 uint64_t seed, int64_t chain0, float *values) {
    const int V = c->V, H1 = c->n[0], H2 = c->n[1];
    float *x = (float *)malloc((size_t)R * H1 * sizeof(float)), *xn = (float *)malloc((size_t)R * H1 * sizeof(float));
    float *v = (float *)malloc((size_t)R * V * sizeof(float)), *h2 = (float *)malloc((size_t)R * H2 * sizeof(float));
    double *lz = (double *)calloc(R, sizeof(double)), *lp = (double *)malloc(R * sizeof(double));
    float *Wt0 = transpose(s->W[0], V, H1), *Wt1 = transpose(s->W[1], H1, H2);
    const orc_key k0 = make_key(seed, SITE_AIS_X0, 0);
    for (int r = 0; r < R; ++r)
        for (int h = 0; h < H1; ++h)
            x[(size_t)r * H1 + h] = (uniform_at(k0, (uint64_t)(chain0 + r) * (uint64_t)H1 + h) < 0.5f) ? 1.0f : 0.0f;
    const float db = 1.0f / (float)n_betas;
#define AIS_TRANSIT(BETA, STEP)                                                                             \
    for (int t = 0; t < k; ++t) {                                                                           \
        orc_act2(x, H1, Wt0, NULL, 0, NULL, V, R, s->vb, s->sigma, (BETA), (BETA), bkind(c),               \
                 c->sample_v, NULL, v, seed, SITE_DBM_V + 16u * (uint32_t)t, (STEP), chain0);               \
        orc_act2(x, H1, s->W[1], NULL, 0, NULL, H2, R, s->hb[1], NULL, (BETA), (BETA), bkind(c),           \
                 c->sample_h[1], NULL, h2, seed, SITE_DBM_H + 1 + 16u * (uint32_t)t, (STEP), chain0);       \
        orc_act2(v, V, s->W[0], h2, H2, Wt1, H1, R, s->hb[0], NULL, (BETA), (BETA), bkind(c),              \
                 c->sample_h[0], NULL, xn, seed, SITE_DBM_H + 0 + 16u * (uint32_t)t, (STEP), chain0);       \
        float *tx = x; x = xn; xn = tx;                                                                     \
    }
    AIS_TRANSIT(db, 0u)                                                     /* x_1 ~ T_1(x_1|x_0)      :704-705 */
    ais_log_p(c, s, x, R, 0.0f, lp);                                        /* -log p_0(x_1)           :708 */
    for (int r = 0; r < R; ++r) lz[r] -= lp[r];
    float beta = db; uint32_t step = 1;
    while (beta < 1.0f - db + 1e-5f) {                                      /* :710-726 */
        ais_log_p(c, s, x, R, beta, lp);
        for (int r = 0; r < R; ++r) lz[r] += lp[r];
        AIS_TRANSIT(beta + db, step)
        ++step;
        ais_log_p(c, s, x, R, beta, lp);
        for (int r = 0; r < R; ++r) lz[r] -= lp[r];
        beta = beta + db;
    }
    ais_log_p(c, s, x, R, 1.0f, lp);                                        /* + log p_M(x_M)          :728 */
    const double logZ0 = (double)(V + H1 + H2) * (double)logf(2.0f);        /* :731-734 */
    for (int r = 0; r < R; ++r) values[r] = (float)(lz[r] + lp[r] + logZ0);
    free(x); free(xn); free(v); free(h2); free(lz); free(lp); free(Wt0); free(Wt1);
#undef AIS_TRANSIT
}

/* log p*_beta(x) with the reference graph's float32 STRUCTURE (dbm.py:650-660): T1 = x.hb0; T1 *= beta; log_p = T1;
 * log_p += reduce_sum(softplus(beta (x W0^T + vb))); log_p += reduce_sum(softplus(beta (x W1 + hb1))) - three float32
 * values combined in float32 (each row sum itself is taken in double and rounded once: the order inside tf.reduce_sum
 * is the backend's). */
static void ais_log_p_f32(const orc_dbm_cfg *c, const orc_dbm_state *s, const float *x, int R, float beta, float *out) {
    const int V = c->V, H1 = c->n[0], H2 = c->n[1];
#pragma omp parallel for schedule(static)
    for (int r = 0; r < R; ++r) {
        const float *xr = x + (size_t)r * H1;
        double t1 = 0.0, sv = 0.0, sh = 0.0;
        for (int h = 0; h < H1; ++h) t1 += (double)xr[h] * (double)s->hb[0][h];
        for (int v = 0; v < V; ++v) {
            double z = s->vb[v];
            for (int h = 0; h < H1; ++h) z += (double)xr[h] * (double)s->W[0][(size_t)v * H1 + h];
            sv += softplus_d(z * (double)beta);
        }
        for (int k2 = 0; k2 < H2; ++k2) {
            double z = s->hb[1][k2];
            for (int h = 0; h < H1; ++h) z += (double)xr[h] * (double)s->W[1][(size_t)h * H2 + k2];
            sh += softplus_d(z * (double)beta);
        }
        float lp = (float)t1 * beta;
        lp = lp + (float)sv;
        lp = lp + (float)sh;
        out[r] = lp;
    }
}

/* AIS with the reference's LITERAL float32 accumulation (dbm.py:708-728): log_Z = -log p_0(x_1); per beta:
 * log_Z += log p_beta(x); x' ~ T(x); log_Z -= log p_beta(x'); finally += log p_1(x_M), += log Z_0 - every operation
 * in float32, in that order.  Same chains (same RNG addressing) as orc_dbm_ais. */
void orc_dbm_ais_literal(const orc_dbm_cfg *c, const orc_dbm_state *s, int n_betas, int R, int k,
                         uint64_t seed, int64_t chain0, float *values) {
    const int V = c->V, H1 = c->n[0], H2 = c->n[1];
    float *x = (float *)malloc((size_t)R * H1 * sizeof(float)), *xn = (float *)malloc((size_t)R * H1 * sizeof(float));
    float *v = (float *)malloc((size_t)R * V * sizeof(float)), *h2 = (float *)malloc((size_t)R * H2 * sizeof(float));
    float *lz = (float *)calloc(R, sizeof(float)), *lp = (float *)malloc(R * sizeof(float));
    float *Wt0 = transpose(s->W[0], V, H1), *Wt1 = transpose(s->W[1], H1, H2);
    const orc_key k0 = make_key(seed, SITE_AIS_X0, 0);
    for (int r = 0; r < R; ++r)
        for (int h = 0; h < H1; ++h)
            x[(size_t)r * H1 + h] = (uniform_at(k0, (uint64_t)(chain0 + r) * (uint64_t)H1 + h) < 0.5f) ? 1.0f : 0.0f;
    const float db = 1.0f / (float)n_betas;
#define AIS_TRANSIT_L(BETA, STEP)                                                                           \
    for (int t = 0; t < k; ++t) {                                                                           \
        orc_act2(x, H1, Wt0, NULL, 0, NULL, V, R, s->vb, s->sigma, (BETA), (BETA), bkind(c),               \
                 c->sample_v, NULL, v, seed, SITE_DBM_V + 16u * (uint32_t)t, (STEP), chain0);               \
        orc_act2(x, H1, s->W[1], NULL, 0, NULL, H2, R, s->hb[1], NULL, (BETA), (BETA), bkind(c),           \
                 c->sample_h[1], NULL, h2, seed, SITE_DBM_H + 1 + 16u * (uint32_t)t, (STEP), chain0);       \
        orc_act2(v, V, s->W[0], h2, H2, Wt1, H1, R, s->hb[0], NULL, (BETA), (BETA), bkind(c),              \
                 c->sample_h[0], NULL, xn, seed, SITE_DBM_H + 0 + 16u * (uint32_t)t, (STEP), chain0);       \
        float *tx = x; x = xn; xn = tx;                                                                     \
    }
    AIS_TRANSIT_L(db, 0u)
    ais_log_p_f32(c, s, x, R, 0.0f, lp);
    for (int r = 0; r < R; ++r) lz[r] = lz[r] - lp[r];
    float beta = db; uint32_t step = 1;
    while (beta < 1.0f - db + 1e-5f) {
        ais_log_p_f32(c, s, x, R, beta, lp);
        for (int r = 0; r < R; ++r) lz[r] = lz[r] + lp[r];
        AIS_TRANSIT_L(beta + db, step)
        ++step;
        ais_log_p_f32(c, s, x, R, beta, lp);
        for (int r = 0; r < R; ++r) lz[r] = lz[r] - lp[r];
        beta = beta + db;
    }
    ais_log_p_f32(c, s, x, R, 1.0f, lp);
    const float logZ0 = (float)(V + H1 + H2) * logf(2.0f);
    for (int r = 0; r < R; ++r) values[r] = (lz[r] + lp[r]) + logZ0;
    free(x); free(xn); free(v); free(h2); free(lz); free(lp); free(Wt0); free(Wt1);
#undef AIS_TRANSIT_L
}

/* variational lower bound terms per row (dbm.py:738-759) given mu from mean-field, double accumulation */
void orc_dbm_log_proba(const orc_dbm_cfg *c, orc_dbm_state *s, const float *X, float *out) {
    orc_dbm_mean_field(c, s, X);
    const int V = c->V, H1 = c->n[0], H2 = c->n[1], N = c->N;
    for (int r = 0; r < N; ++r) {
        const float *x = X + (size_t)r * V, *m0 = s->mu[0] + (size_t)r * H1, *m1 = s->mu[1] + (size_t)r * H2;
        double e = 0.0;
        for (int h = 0; h < H1; ++h) {
            double z = 0.0;
            for (int v = 0; v < V; ++v) z += (double)x[v] * (double)s->W[0][(size_t)v * H1 + h];
            e += z * (double)m0[h];
        }
        for (int k2 = 0; k2 < H2; ++k2) {
            double z = 0.0;
            for (int h = 0; h < H1; ++h) z += (double)m0[h] * (double)s->W[1][(size_t)h * H2 + k2];
            e += z * (double)m1[k2];
        }
        for (int v = 0; v < V; ++v) e += (double)x[v] * (double)s->vb[v];
        for (int h = 0; h < H1; ++h) e += (double)m0[h] * (double)s->hb[0][h];
        for (int k2 = 0; k2 < H2; ++k2) e += (double)m1[k2] * (double)s->hb[1][k2];
        double ent = 0.0;
        for (int h = 0; h < H1; ++h) { double q = fmin(fmax((double)m0[h], 1e-7), 1.0 - 1e-7); ent += -q * log(q) - (1 - q) * log(1 - q); }
        for (int k2 = 0; k2 < H2; ++k2) { double q = fmin(fmax((double)m1[k2], 1e-7), 1.0 - 1e-7); ent += -q * log(q) - (1 - q) * log(1 - q); }
        out[r] = (float)(e + ent);
    }
}

/* ======================================================================= float64 RBM path
 * The reference's dtype is a constructor argument (base/mixin.py:15) and its own tests train a
 * float64 BernoulliRBM (rbm/tests/test_rbm.py:53-56,70-73).  Same graph as the float32 functions
 * above (base_rbm.py:415-531), every operation in IEEE double; canonical order unchanged.
 * RNG: TF draws a float64 uniform from a word pair (Uint64ToDouble), 2 per Philox block. */
static double u64_to_uniform_d(uint32_t x0, uint32_t x1) {
    union { uint64_t u; double d; } v;
    v.u = ((uint64_t)1023 << 52) | (((uint64_t)x0 & 0xfffffu) << 32) | (uint64_t)x1;
    return v.d - 1.0;
}
static double uniform_at_d(orc_key key, uint64_t idx) {
    uint32_t w[4];
    philox_block(key, idx >> 1, w);
    return (idx & 1) ? u64_to_uniform_d(w[2], w[3]) : u64_to_uniform_d(w[0], w[1]);
}
static double normal_at_d(orc_key key, uint64_t idx) {      /* BoxMullerDouble: one pair per block */
    uint32_t w[4];
    philox_block(key, idx >> 1, w);
    double u1 = u64_to_uniform_d(w[0], w[1]);
    if (u1 < 1.0e-20) u1 = 1.0e-20;
    const double v1 = 6.283185307179586476925286766559 * u64_to_uniform_d(w[2], w[3]);
    const double r = sqrt(-2.0 * log(u1));
    return (idx & 1) ? cos(v1) * r : sin(v1) * r;
}
void orc_uniform_d(uint64_t seed, uint32_t site, uint32_t call, uint64_t idx0, uint64_t n, double *out) {
    orc_key k = make_key(seed, site, call);
    for (uint64_t i = 0; i < n; ++i) out[i] = uniform_at_d(k, idx0 + i);
}

/* exp(-a), a in [0, 700]: Cody-Waite + degree-13 Taylor in Horner form, fma only */
static double exp_neg_d(double a) {
    const double t = a * -1.4426950408889634074;
    const double n = rint(t);
    double r = fma(n, -6.93147180369123816490e-01, -a);      /* ln2_hi */
    r = fma(n, -1.90821492927058770002e-10, r);               /* ln2_lo */
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
    union { uint64_t u; double d; } v;
    v.d = p;
    v.u += ((uint64_t)(int64_t)n) << 52;
    return v.d;
}
double orc_sigmoid_d(double x) {
    double a = fabs(x);
    if (a > 700.0) a = 700.0;
    const double e = exp_neg_d(a);
    const double d = 1.0 + e;
    return (x >= 0.0) ? (1.0 / d) : (e / d);
}

typedef struct { double *W, *vb, *hb, *dW, *dvb, *dhb, *q, *sigma; } orc_rbm_state_d;
typedef struct { double *Xin, *h0m, *h0s, *vm, *vs, *hm, *hs; } orc_rbm_work_d;

/* MultinomialLayer in double (the float64 twin of softmax_multinomial_row): same operation sequence with
 * exp_neg_d (argument clamped at 700), sequential prefix sums, uniforms from uniform_at_d */
static double exp_neg_d(double a);
static void softmax_multinomial_row_d(const double *l, int I, int M, int sample, double *means, double *states,
                                      orc_key key, uint64_t row, double *e, double *c) {
    double mx = l[0];
    for (int i = 1; i < I; ++i) mx = fmax(mx, l[i]);
    double run = 0.0;
    for (int i = 0; i < I; ++i) {
        double a = mx - l[i];
        if (a > 700.0) a = 700.0;
        e[i] = exp_neg_d(a);
        run = run + e[i];
        c[i] = run;
    }
    const double S = c[I - 1];
    for (int i = 0; i < I; ++i) {
        const double m = (double)M * (e[i] / S);
        if (means) means[i] = m;
        if (states) states[i] = sample ? 0.0 : m;
    }
    if (sample && states) {
        for (int d = 0; d < M; ++d) {
            const double u = uniform_at_d(key, row * (uint64_t)M + (uint64_t)d);
            const double t = u * S;
            int lo = 0, hi = I - 1;                 /* smallest i with c[i] > t */
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (c[mid] > t) hi = mid; else lo = mid + 1;
            }
            states[lo] = states[lo] + 1.0;
        }
    }
}

/* z[j][i] = sum_k Q[j][k] * Pk[k][i], then the layer activation / draw (layers.py:34-36,47-51,84-89);
 * kind >= 16: MultinomialLayer with kind - 16 samples (layers.py:54-70) */
static void act_d(const double *Q, int K, const double *Pk, int I, int J, const double *bias, const double *sigma,
                  double mult, int kind, int sample, double *means, double *states,
                  uint64_t seed, uint32_t site, uint32_t call, int64_t row0) {
    const orc_key key = make_key(seed, site, call);
#pragma omp parallel
    {
        double *acc = (double *)malloc((size_t)I * sizeof(double));
#pragma omp for schedule(static)
        for (int j = 0; j < J; ++j) {
            for (int i = 0; i < I; ++i) acc[i] = 0.0;
            for (int k = 0; k < K; ++k) {
                const double q = Q[(size_t)j * K + k];
                const double *p = Pk + (size_t)k * I;
                for (int i = 0; i < I; ++i) acc[i] = fma(p[i], q, acc[i]);
            }
            if (kind >= 16) {
                double *e = (double *)malloc(2 * (size_t)I * sizeof(double));
                for (int i = 0; i < I; ++i) acc[i] = mult * acc[i] + mult * bias[i];
                softmax_multinomial_row_d(acc, I, kind - 16, sample, means ? means + (size_t)j * I : NULL,
                                          states ? states + (size_t)j * I : NULL, key, (uint64_t)(row0 + j), e, e + I);
                free(e);
                continue;
            }
            for (int i = 0; i < I; ++i) {
                const double x = mult * acc[i];
                const double b = mult * bias[i];
                const double m = (kind == UNIT_BERNOULLI) ? orc_sigmoid_d(x + b) : (x * sigma[i] + b);
                double s = m;
                if (sample) {
                    const uint64_t idx = (uint64_t)(row0 + j) * (uint64_t)I + (uint64_t)i;
                    if (kind == UNIT_BERNOULLI) s = (uniform_at_d(key, idx) < m) ? 1.0 : 0.0;
                    else s = normal_at_d(key, idx) * sigma[i] + m;
                }
                if (means) means[(size_t)j * I + i] = m;
                if (states) states[(size_t)j * I + i] = s;
            }
        }
        free(acc);
    }
}

/* hy = {l2, sparsity_target, sparsity_cost, sparsity_damping, dropout (<0: off)} as doubles */
void orc_rbm_chain_d(const orc_rbm_cfg *c, const double *hy, const orc_rbm_state_d *s, const double *X, int B, int k,
                     uint64_t seed, uint32_t call, int64_t row0, orc_rbm_work_d *w) {
    const int V = c->V, H = c->H;
    const size_t nX = (size_t)B * V;
    for (size_t e = 0; e < nX; ++e) {
        double x = X[e];
        if (c->v_unit == UNIT_GAUSSIAN) x = x / s->sigma[e % (size_t)V];
        w->Xin[e] = x;
    }
    if (hy[4] >= 0.0) {
        const orc_key key = make_key(seed, SITE_DROPOUT, call);
        const double keep = hy[4];
        for (size_t e = 0; e < nX; ++e) {
            const double u = uniform_at_d(key, (uint64_t)row0 * (uint64_t)V + e);
            w->Xin[e] = (w->Xin[e] / keep) * floor(keep + u);
        }
    }
    const double up = 1.0 + (c->dbm_first ? 1.0 : 0.0), down = 1.0 + (c->dbm_last ? 1.0 : 0.0);
    double *Wt = (double *)malloc((size_t)V * H * sizeof(double));
    for (int v = 0; v < V; ++v) for (int h = 0; h < H; ++h) Wt[(size_t)h * V + v] = s->W[(size_t)v * H + h];
    const int hkind = (c->h_unit == UNIT_MULTINOMIAL) ? 16 + c->n_samples : UNIT_BERNOULLI;
    act_d(w->Xin, V, s->W, H, B, s->hb, NULL, up, hkind, 1, w->h0m, w->h0s, seed, SITE_H0, call, row0);
    const double *hstate = c->sample_h ? w->h0s : w->h0m;
    for (int t = 0; t < k; ++t) {
        act_d(hstate, H, Wt, V, B, s->vb, s->sigma, down, c->v_unit, c->sample_v, w->vm, w->vs,
              seed, SITE_V + 16u * (uint32_t)t, call, row0);
        act_d(w->vs, V, s->W, H, B, s->hb, NULL, up, hkind, c->sample_h, w->hm, w->hs,
              seed, SITE_H + 16u * (uint32_t)t, call, row0);
        hstate = w->hs;
    }
    free(Wt);
}

/* chain + raw sums + update (base_rbm.py:443-478), one call = session.run(train_op) */
void orc_rbm_train_step_d(const orc_rbm_cfg *c, const double *hy, orc_rbm_state_d *s, const double *X, int B,
                          double lr, double mom, int k, uint64_t seed, uint32_t call, int64_t row0,
                          orc_rbm_work_d *w) {
    const int V = c->V, H = c->H;
    orc_rbm_chain_d(c, hy, s, X, B, k, seed, call, row0, w);
    const double N = (double)B, l2 = hy[0];
    double *sv = (double *)calloc((size_t)V + 2 * (size_t)H, sizeof(double)), *sh = sv + V, *sq = sh + H;
    for (int b = 0; b < B; ++b) {
        for (int v = 0; v < V; ++v) sv[v] = sv[v] + (w->Xin[(size_t)b * V + v] - w->vs[(size_t)b * V + v]);
        for (int h = 0; h < H; ++h) {
            sh[h] = sh[h] + (w->h0m[(size_t)b * H + h] - w->hm[(size_t)b * H + h]);
            sq[h] = sq[h] + w->hm[(size_t)b * H + h];
        }
    }
    double *pen = (double *)malloc((size_t)H * sizeof(double));
    for (int v = 0; v < V; ++v) {
        const double g = sv[v] / N;
        const double d = lr * (mom * s->dvb[v] + g);
        s->dvb[v] = d;
        s->vb[v] = s->vb[v] + d;
    }
    const double damp = hy[3], cost = hy[2], target = hy[1];
    for (int h = 0; h < H; ++h) {
        const double qn = damp * s->q[h] + (1.0 - damp) * sq[h];
        s->q[h] = qn;
        pen[h] = cost * (qn - target);
        double g = sh[h] / N;
        g = g - pen[h];
        const double d = lr * (mom * s->dhb[h] + g);
        s->dhb[h] = d;
        s->hb[h] = s->hb[h] + d;
    }
#pragma omp parallel for schedule(static)
    for (int v = 0; v < V; ++v) {
        for (int h = 0; h < H; ++h) {
            double acc = 0.0;                          /* one chain: positive rows, then negated negative rows */
            for (int b = 0; b < B; ++b) acc = fma(w->h0m[(size_t)b * H + h], w->Xin[(size_t)b * V + v], acc);
            for (int b = 0; b < B; ++b) acc = fma(w->hm[(size_t)b * H + h], -w->vs[(size_t)b * V + v], acc);
            const size_t e = (size_t)v * H + h;
            double g = acc / N;
            g = g - l2 * s->W[e];
            g = g - pen[h];
            const double d = lr * (mom * s->dW[e] + g);
            s->dW[e] = d;
            s->W[e] = s->W[e] + d;
        }
    }
    free(pen); free(sv);
}

/* h_hat of the MultinomialRBM free energy in double: counts of M uniform category draws floor(u * K) */
static void multinomial_uniform_counts_d(int K, int M, uint64_t seed, uint32_t call, uint32_t t, double *hhat) {
    const orc_key key = make_key(seed, SITE_FE + 16u * t, call);
    for (int k = 0; k < K; ++k) hhat[k] = 0.0;
    for (int d = 0; d < M; ++d) {
        int idx = (int)(uniform_at_d(key, (uint64_t)d) * (double)K);
        if (idx > K - 1) idx = K - 1;
        hhat[idx] += 1.0;
    }
}

double orc_rbm_free_energy_d_ex(const orc_rbm_cfg *c, const orc_rbm_state_d *s, const double *Xin, int B,
                                const int32_t *flip, uint64_t seed, uint32_t call, uint32_t t_stream) {
    const int V = c->V, H = c->H;
    double total = 0.0;
    double *hhat = NULL;
    if (c->h_unit == UNIT_MULTINOMIAL) {
        hhat = (double *)malloc((size_t)H * sizeof(double));
        multinomial_uniform_counts_d(H, c->n_samples, seed, call, t_stream, hhat);
    }
    for (int b = 0; b < B; ++b) {
        const double *x = Xin + (size_t)b * V;
        double t = 0.0;
        for (int v = 0; v < V; ++v) {
            double xv = x[v];
            if (flip && flip[b] == v) xv = 1.0 - xv;
            if (c->v_unit == UNIT_GAUSSIAN) { const double mu = s->vb[v] / s->sigma[v]; t += 0.5 * (xv - mu) * (xv - mu); }
            else t -= xv * s->vb[v];
        }
        for (int h = 0; h < H; ++h) {
            double z = hhat ? 0.0 : s->hb[h];
            for (int v = 0; v < V; ++v) {
                double xv = x[v];
                if (flip && flip[b] == v) xv = 1.0 - xv;
                z += xv * s->W[(size_t)v * H + h];
            }
            t -= hhat ? z * hhat[h] : softplus_d(z);                      /* rbm.py:57-60 */
        }
        total += t;
    }
    double fe = total / B;
    if (hhat) {
        const double M = c->n_samples, K = H;
        fe += -lgamma(M + K) + lgamma(M + 1.0) + lgamma(K);                /* rbm.py:61 */
        free(hhat);
    }
    return fe;
}
double orc_rbm_free_energy_d(const orc_rbm_cfg *c, const orc_rbm_state_d *s, const double *Xin, int B,
                             const int32_t *flip) {
    return orc_rbm_free_energy_d_ex(c, s, Xin, B, flip, 0, 0, 0);
}

/* out = [msre, pll, l2_loss, free_energy] from a finished chain (base_rbm.py:482-517) */
void orc_rbm_metrics_d(const orc_rbm_cfg *c, const double *hy, const orc_rbm_state_d *s, const orc_rbm_work_d *w,
                       int B, uint64_t seed, uint32_t call, int64_t row0, double *out4) {
    const int V = c->V, H = c->H;
    double se = 0.0, l2 = 0.0;
    for (size_t e = 0; e < (size_t)B * V; ++e) { const double d = w->Xin[e] - w->vm[e]; se += d * d; }
    for (size_t e = 0; e < (size_t)V * H; ++e) l2 += s->W[e] * s->W[e];
    out4[0] = se / ((double)B * V);
    out4[2] = hy[0] * (0.5 * l2);
    int32_t *flip = (int32_t *)malloc((size_t)B * sizeof(int32_t));
    const orc_key key = make_key(seed, SITE_PLL, call);
    for (int b = 0; b < B; ++b) {
        const uint64_t idx = (uint64_t)row0 + (uint64_t)b;
        uint32_t wd[4];
        philox_block(key, idx >> 2, wd);
        flip[b] = (int32_t)(wd[idx & 3] % (uint32_t)V);
    }
    const int mn = c->h_unit == UNIT_MULTINOMIAL;     /* a fresh h_hat per _free_energy() call: streams t = 0, 1, 2 */
    const double fe = orc_rbm_free_energy_d_ex(c, s, w->Xin, B, NULL, seed, call, 0);
    const double fe1 = mn ? orc_rbm_free_energy_d_ex(c, s, w->Xin, B, NULL, seed, call, 1) : fe;
    const double fe2 = orc_rbm_free_energy_d_ex(c, s, w->Xin, B, flip, seed, call, 2);
    out4[1] = (double)V * -softplus_d(-(fe2 - fe1));
    out4[3] = fe;
    free(flip);
}

/* the DBM path in float64 (shares the helpers above) */
#include "bm_oracle_dbm64.c"
