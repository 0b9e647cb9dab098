/* bm_oracle_dbm64.c - the DBM functions of bm_oracle.c in IEEE double (included at the end of bm_oracle.c: it shares the
 * Philox / sigmoid / softmax helpers of the float64 RBM path).  TEST INFRASTRUCTURE ONLY, like the rest of oracle/.
 *
 * The reference's dtype is a constructor argument (base/mixin.py:14-25) and the DBM graph is built "all in model dtype"
 * (dbm.py:294-383): with dtype='float64' every tensor, every hyper-parameter placeholder and every random draw is float64.
 * This file restates dbm.py:385-759 for that dtype.  It is the float32 section of bm_oracle.c (`orc_dbm_*`) converted
 * MECHANICALLY - float -> double, the f-suffixed literals and libm calls, `_d` names - so that the two cannot drift apart;
 * only the contraction helpers differ by design: every dot product / outer product / column sum of the float64 path is the
 * SEQUENTIAL ascending-k fma chain (what v_mfma_f64_16x16x4_f64 computes when the chain is continued through the
 * accumulator, csrc/bm_rbm64.hip), where the float32 path uses the blocked canonical order of the fp32 tile engine. */

static double *transpose_d(const double *A, int R, int C) {   /* A[R][C] -> T[C][R] */
    double *T = (double *)malloc((size_t)R * C * sizeof(double));
    for (int r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) T[(size_t)c * R + r] = A[(size_t)r * C + c];
    return T;
}

/* out[j][i] = sum_b sgn * Qb[b][j] * Pb[b][i], rows b ascending (one fma chain per output) */
static void outer_chain_d(double *out, const double *Qb, int J, const double *Pb, int I, int B, double sgn, int accumulate) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < J; ++j) {
        double *acc = out + (size_t)j * I;
        if (!accumulate) for (int i = 0; i < I; ++i) acc[i] = 0.0;
        for (int b = 0; b < B; ++b) {
            const double q = sgn * Qb[(size_t)b * J + j];
            const double *p = Pb + (size_t)b * I;
            for (int i = 0; i < I; ++i) acc[i] = fma(p[i], q, acc[i]);
        }
    }
}

static void colsum_diff_d(double *out, const double *A, const double *Bm, int B, int C) {
    for (int c = 0; c < C; ++c) out[c] = 0.0;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            double x = A[(size_t)b * C + c];
            if (Bm) x = x - Bm[(size_t)b * C + c];
            out[c] = out[c] + x;
        }
}

/* act_d with an optional second K segment and a separate bias multiplier (the fused stage of the DBM passes) */
void orc_act2_d(const double *Q1, int K1, const double *P1k, const double *Q2, int K2, const double *P2k,
                int I, int J, const double *bias, const double *sigma, double mult, double bmult, int kind, int sample,
                double *means, double *states, uint64_t seed, uint32_t site, uint32_t call, int64_t row0) {
    const orc_key key = make_key(seed, site, call);
#pragma omp parallel
    {
        double *acc = (double *)malloc((size_t)I * sizeof(double));
#pragma omp for schedule(static)
        for (int j = 0; j < J; ++j) {
            for (int i = 0; i < I; ++i) acc[i] = 0.0;
            for (int k = 0; k < K1; ++k) {
                const double q = Q1[(size_t)j * K1 + k];
                const double *p = P1k + (size_t)k * I;
                for (int i = 0; i < I; ++i) acc[i] = fma(p[i], q, acc[i]);
            }
            for (int k = 0; k < K2; ++k) {
                const double q = Q2[(size_t)j * K2 + k];
                const double *p = P2k + (size_t)k * I;
                for (int i = 0; i < I; ++i) acc[i] = fma(p[i], q, acc[i]);
            }
            if (kind >= 16) {
                double *e = (double *)malloc(2 * (size_t)I * sizeof(double));
                for (int i = 0; i < I; ++i) acc[i] = mult * acc[i] + bmult * bias[i];
                softmax_multinomial_row_d(acc, I, kind - 16, sample, means ? means + (size_t)j * I : NULL,
                                          states ? states + (size_t)j * I : NULL, key, (uint64_t)(row0 + j), e, e + I);
                free(e);
                continue;
            }
            for (int i = 0; i < I; ++i) {
                const double x = mult * acc[i];
                const double b = bmult * bias[i];
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

/* ---- from here on: the float32 DBM section of bm_oracle.c, converted mechanically (see the header) ---- */
typedef struct {
    int32_t L, V, n[ORC_MAXL];            /* hidden layer sizes */
    int32_t v_unit, sample_v, sample_h[ORC_MAXL];
    int32_t N, M, max_mf;
    double mf_tol, l2, max_norm, sp_target[ORC_MAXL], sp_cost[ORC_MAXL], sp_damping;
    int32_t h_unit[ORC_MAXL], n_samples[ORC_MAXL];   /* hidden layer kinds (layers.py:39-70): Bernoulli | Multinomial(n_samples) */
} orc_dbm_cfg_d;

/* activation kind of hidden layer i for orc_act2 (Multinomial: 16 + n_samples) */
static int hkind_d(const orc_dbm_cfg_d *c, int i) {
    return (c->h_unit[i] == UNIT_MULTINOMIAL) ? 16 + c->n_samples[i] : UNIT_BERNOULLI;
}
/* the same for the visible layer / an all-Bernoulli stack (AIS) */
static int vkind_d(const orc_dbm_cfg_d *c) {
    return c->v_unit;
}
static int bkind_d(const orc_dbm_cfg_d *c) { (void)c; return UNIT_BERNOULLI; }

typedef struct {
    double *W[ORC_MAXL], *dW[ORC_MAXL], *hb[ORC_MAXL], *dhb[ORC_MAXL], *q[ORC_MAXL], *mm[ORC_MAXL];
    double *mu[ORC_MAXL], *mu_new[ORC_MAXL], *H[ORC_MAXL], *H_new[ORC_MAXL];
    double *vb, *dvb, *sigma, *v, *v_new;
    double *wnorm[ORC_MAXL];
} orc_dbm_state_d;


static int dn_d(const orc_dbm_cfg_d *c, int i) { return i == 0 ? c->V : c->n[i - 1]; }   /* n[0]=V, n[i+1]=hidden i */

/* `_make_gibbs_step` (dbm.py:385-427): bottom-up sweep with NEW below / OLD above */
static void dbm_sweep_d(const orc_dbm_cfg_d *c, const orc_dbm_state_d *s, int J, const double *vin, double *const *Hin,
                      double *vout, double *const *Hout, int update_v, int sample, int t,
                      uint64_t seed, uint32_t call, int64_t row0) {
    const int L = c->L;
    for (int i = 0; i < L; ++i) {
        const double *below = (i == 0) ? vin : Hout[i - 1];
        const int Kb = dn_d(c, i), I = dn_d(c, i + 1);
        double *Wt = NULL; const double *above = NULL; int Ka = 0;
        if (i + 1 < L) { Ka = dn_d(c, i + 2); Wt = transpose_d(s->W[i + 1], I, Ka); above = Hin[i + 1]; }
        const int smp = sample && c->sample_h[i];
        orc_act2_d(below, Kb, s->W[i], above, Ka, Wt, I, J, s->hb[i], NULL, 1.0, 1.0, hkind_d(c, i), smp,
                 NULL, Hout[i], seed, SITE_DBM_H + (uint32_t)i + 16u * (uint32_t)t, call, row0);
        free(Wt);
    }
    if (update_v) {
        double *Wt0 = transpose_d(s->W[0], c->V, dn_d(c, 1));
        const int smp = sample && c->sample_v;
        orc_act2_d(Hout[0], dn_d(c, 1), Wt0, NULL, 0, NULL, c->V, J, s->vb, s->sigma, 1.0, 1.0, vkind_d(c), smp,
                 NULL, vout, seed, SITE_DBM_V + 16u * (uint32_t)t, call, row0);
        free(Wt0);
    }
}

static double max_abs_diff_d(const double *a, const double *b, size_t n) {
    double m = 0.0;
    for (size_t e = 0; e < n; ++e) { const double d = fabs(a[e] - b[e]); if (d > m) m = d; }
    return m;
}

/* `_make_mf` (dbm.py:429-478); result in s->mu, init values in s->mu_new; returns the sweeps run */
int orc_dbm_mean_field_d(const orc_dbm_cfg_d *c, orc_dbm_state_d *s, const double *X) {
    const int L = c->L, N = c->N;
    for (int i = 0; i < L; ++i) {                              /* approx-inference init :434-446 */
        const double *below = (i == 0) ? X : s->mu_new[i - 1];
        const double mult = (i == 0 || i < L - 1) ? 2.0 : 1.0;
        orc_act2_d(below, dn_d(c, i), s->W[i], NULL, 0, NULL, dn_d(c, i + 1), N, s->hb[i], NULL, mult, 1.0,
                 hkind_d(c, i), 0, s->mu_new[i], NULL, 0, 0, 0, 0);
    }
    double diff = 0.0;
    for (int i = 0; i < L; ++i) {
        const double d = max_abs_diff_d(s->mu[i], s->mu_new[i], (size_t)N * dn_d(c, i + 1));
        if (d > diff) diff = d;
    }
    double *cur[ORC_MAXL], *alt[ORC_MAXL];
    for (int i = 0; i < L; ++i) {
        cur[i] = s->mu[i];
        alt[i] = (double *)malloc((size_t)N * dn_d(c, i + 1) * sizeof(double));
    }
    int step = 0;
    while (step < c->max_mf && diff > c->mf_tol) {             /* cond :449-452, body :454-457 */
        dbm_sweep_d(c, s, N, X, cur, NULL, alt, 0, 0, 0, 0, 0, 0);
        diff = 0.0;
        for (int i = 0; i < L; ++i) {
            const double d = max_abs_diff_d(cur[i], alt[i], (size_t)N * dn_d(c, i + 1));
            if (d > diff) diff = d;
        }
        for (int i = 0; i < L; ++i) { double *t = cur[i]; cur[i] = alt[i]; alt[i] = t; }
        ++step;
    }
    for (int i = 0; i < L; ++i) {                              /* mu.assign(result) :477 */
        if (cur[i] != s->mu[i]) { memcpy(s->mu[i], cur[i], (size_t)N * dn_d(c, i + 1) * sizeof(double)); free(cur[i]); }
        else free(alt[i]);
    }
    return step;
}

/* `_make_particles_update` (dbm.py:480-509): k sweeps with swap; latest state ends in v/H */
void orc_dbm_particles_d(const orc_dbm_cfg_d *c, orc_dbm_state_d *s, int k, int sample,
                       uint64_t seed, uint32_t call, int64_t prow0) {
    for (int t = 0; t < k; ++t) {
        dbm_sweep_d(c, s, c->M, s->v, s->H, s->v_new, s->H_new, 1, sample, t, seed, call, prow0);
        double *tv = s->v; s->v = s->v_new; s->v_new = tv;
        for (int i = 0; i < c->L; ++i) { double *th = s->H[i]; s->H[i] = s->H_new[i]; s->H_new[i] = th; }
    }
}

/* reconstruction sigma(mu0 W0^T + vb) (dbm.py:625-628) */
void orc_dbm_reconstruct_from_mu_d(const orc_dbm_cfg_d *c, const orc_dbm_state_d *s, double *R) {
    double *Wt0 = transpose_d(s->W[0], c->V, dn_d(c, 1));
    orc_act2_d(s->mu[0], dn_d(c, 1), Wt0, NULL, 0, NULL, c->V, c->N, s->vb, s->sigma, 1.0, 1.0, vkind_d(c), 0,
             R, NULL, 0, 0, 0, 0);
    free(Wt0);
}

/* gradients + sparsity + momentum + max-norm (dbm.py:550-621) */
static void dbm_apply_update_d(const orc_dbm_cfg_d *c, orc_dbm_state_d *s, const double *X, double lr, double mom) {
    const int L = c->L;
    const double N = (double)c->N, M = (double)c->M;
    double *sx = (double *)malloc(c->V * sizeof(double)), *sv = (double *)malloc(c->V * sizeof(double));
    colsum_diff_d(sx, X, NULL, c->N, c->V);
    colsum_diff_d(sv, s->v, NULL, c->M, c->V);
    double *pen[ORC_MAXL];
    /* the W update reads the gradients of the PRE-update state: compute all raw sums first */
    double *pos[ORC_MAXL], *neg[ORC_MAXL];
    for (int i = 0; i < L; ++i) {
        const int J = dn_d(c, i), I = dn_d(c, i + 1);
        pos[i] = (double *)malloc((size_t)J * I * sizeof(double));
        neg[i] = (double *)malloc((size_t)J * I * sizeof(double));
        outer_chain_d(pos[i], (i == 0) ? X : s->mu[i - 1], J, s->mu[i], I, c->N, 1.0, 0);      /* :556,565 */
        outer_chain_d(neg[i], (i == 0) ? s->v : s->H[i - 1], J, s->H[i], I, c->M, 1.0, 0);    /* :557,566 */
    }
    for (int v = 0; v < c->V; ++v) {                                                          /* :553, :597-600 */
        const double g = sx[v] / N - sv[v] / M;
        const double d = lr * (mom * s->dvb[v] + g);
        s->dvb[v] = d;
        s->vb[v] = s->vb[v] + d;
    }
    for (int i = 0; i < L; ++i) {
        const int n = dn_d(c, i + 1);
        double *smu = (double *)malloc(n * sizeof(double)), *sH = (double *)malloc(n * sizeof(double));
        colsum_diff_d(smu, s->mu[i], NULL, c->N, n);
        colsum_diff_d(sH, s->H[i], NULL, c->M, n);
        pen[i] = (double *)malloc(n * sizeof(double));
        for (int h = 0; h < n; ++h) {
            double g = smu[h] / N - sH[h] / M;                                                 /* :573-576 */
            const double qn = c->sp_damping * s->q[i][h] + (1.0 - c->sp_damping) * sH[i];     /* :582-584 scalar index quirk */
            const double mn = c->sp_damping * s->mm[i][h] + (1.0 - c->sp_damping) * smu[i];   /* :585-587 */
            s->q[i][h] = qn;
            s->mm[i][h] = mn;
            const double p1 = c->sp_cost[i] * (qn - c->sp_target[i]);
            const double p2 = c->sp_cost[i] * (mn - c->sp_target[i]);
            pen[i][h] = p1 + p2;
            g = g - pen[i][h];
            const double d = lr * (mom * s->dhb[i][h] + g);
            s->dhb[i][h] = d;
            s->hb[i][h] = s->hb[i][h] + d;
        }
        free(smu); free(sH);
    }
    for (int i = 0; i < L; ++i) {
        const int J = dn_d(c, i), I = dn_d(c, i + 1);
        for (int j = 0; j < J; ++j)
            for (int h = 0; h < I; ++h) {
                const size_t e = (size_t)j * I + h;
                double g = pos[i][e] / N - neg[i][e] / M;                                      /* :556-558 */
                g = g - c->l2 * s->W[i][e];
                g = g - pen[i][h];                                                            /* :590 */
                const double d = lr * (mom * s->dW[i][e] + g);                                 /* :604 */
                s->dW[i][e] = d;
                s->W[i][e] = s->W[i][e] + d;                                                  /* :605 */
            }
        for (int h = 0; h < I; ++h) {                                                         /* max-norm :511-513,606-607 */
            double acc = 0.0;
            for (int j = 0; j < J; ++j) { const double w = s->W[i][(size_t)j * I + h]; acc = fma(w, w, acc); }
            const double nrm = sqrt(acc);
            const double num = fmin(nrm, c->max_norm), den = fmax(nrm, 1e-8);
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
int orc_dbm_train_step_d(const orc_dbm_cfg_d *c, orc_dbm_state_d *s, const double *X, double lr, double mom, int k,
                       uint64_t seed, uint32_t call, int64_t prow0, double *msre) {
    const int nmf = orc_dbm_mean_field_d(c, s, X);
    orc_dbm_particles_d(c, s, k, 1, seed, call, prow0);
    if (msre) {
        double *R = (double *)malloc((size_t)c->N * c->V * sizeof(double));
        orc_dbm_reconstruct_from_mu_d(c, s, R);
        double se = 0.0;
        for (size_t e = 0; e < (size_t)c->N * c->V; ++e) { const double d = (double)X[e] - (double)R[e]; se += d * d; }
        *msre = (double)(se / ((double)c->N * c->V));
        free(R);
    }
    dbm_apply_update_d(c, s, X, lr, mom);
    return nmf;
}

/* sample_v op (dbm.py:641-648): k sampled sweeps (assigned), then k mean sweeps whose v is assigned */
void orc_dbm_sample_v_d(const orc_dbm_cfg_d *c, orc_dbm_state_d *s, int k, uint64_t seed, uint32_t call, int64_t prow0) {
    orc_dbm_particles_d(c, s, k, 1, seed, call, prow0);
    const int L = c->L, M = c->M;
    double *Hin[ORC_MAXL], *Hout[ORC_MAXL], *vin, *vout;
    vin = (double *)malloc((size_t)M * c->V * sizeof(double)); vout = (double *)malloc((size_t)M * c->V * sizeof(double));
    memcpy(vin, s->v, (size_t)M * c->V * sizeof(double));
    for (int i = 0; i < L; ++i) {
        const size_t n = (size_t)M * dn_d(c, i + 1);
        Hin[i] = (double *)malloc(n * sizeof(double)); Hout[i] = (double *)malloc(n * sizeof(double));
        memcpy(Hin[i], s->H[i], n * sizeof(double));
    }
    for (int t = 0; t < k; ++t) {
        dbm_sweep_d(c, s, M, vin, Hin, vout, Hout, 1, 0, k + t, seed, call, prow0);
        double *tv = vin; vin = vout; vout = tv;
        for (int i = 0; i < L; ++i) { double *th = Hin[i]; Hin[i] = Hout[i]; Hout[i] = th; }
    }
    memcpy(s->v, vin, (size_t)M * c->V * sizeof(double));
    free(vin); free(vout);
    for (int i = 0; i < L; ++i) { free(Hin[i]); free(Hout[i]); }
}

/* log p*_beta(x) per chain (dbm.py:650-660), double accumulation */
static void ais_log_p_d(const orc_dbm_cfg_d *c, const orc_dbm_state_d *s, const double *x, int R, double beta, double *out) {
    const int V = c->V, H1 = c->n[0], H2 = c->n[1];
#pragma omp parallel for schedule(static)
    for (int r = 0; r < R; ++r) {
        const double *xr = x + (size_t)r * H1;
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
void orc_dbm_ais_d(const orc_dbm_cfg_d *c, const orc_dbm_state_d *s, int n_betas, int R, int k,
                 uint64_t seed, int64_t chain0, double *values) {
    const int V = c->V, H1 = c->n[0], H2 = c->n[1];
    double *x = (double *)malloc((size_t)R * H1 * sizeof(double)), *xn = (double *)malloc((size_t)R * H1 * sizeof(double));
    double *v = (double *)malloc((size_t)R * V * sizeof(double)), *h2 = (double *)malloc((size_t)R * H2 * sizeof(double));
    double *lz = (double *)calloc(R, sizeof(double)), *lp = (double *)malloc(R * sizeof(double));
    double *Wt0 = transpose_d(s->W[0], V, H1), *Wt1 = transpose_d(s->W[1], H1, H2);
    const orc_key k0 = make_key(seed, SITE_AIS_X0, 0);
    for (int r = 0; r < R; ++r)
        for (int h = 0; h < H1; ++h)
            x[(size_t)r * H1 + h] = (uniform_at_d(k0, (uint64_t)(chain0 + r) * (uint64_t)H1 + h) < 0.5) ? 1.0 : 0.0;
    const double db = 1.0 / (double)n_betas;
#define AIS_TRANSIT(BETA, STEP)                                                                             \
    for (int t = 0; t < k; ++t) {                                                                           \
        orc_act2_d(x, H1, Wt0, NULL, 0, NULL, V, R, s->vb, s->sigma, (BETA), (BETA), bkind_d(c),               \
                 c->sample_v, NULL, v, seed, SITE_DBM_V + 16u * (uint32_t)t, (STEP), chain0);               \
        orc_act2_d(x, H1, s->W[1], NULL, 0, NULL, H2, R, s->hb[1], NULL, (BETA), (BETA), bkind_d(c),           \
                 c->sample_h[1], NULL, h2, seed, SITE_DBM_H + 1 + 16u * (uint32_t)t, (STEP), chain0);       \
        orc_act2_d(v, V, s->W[0], h2, H2, Wt1, H1, R, s->hb[0], NULL, (BETA), (BETA), bkind_d(c),              \
                 c->sample_h[0], NULL, xn, seed, SITE_DBM_H + 0 + 16u * (uint32_t)t, (STEP), chain0);       \
        double *tx = x; x = xn; xn = tx;                                                                     \
    }
    AIS_TRANSIT(db, 0u)                                                     /* x_1 ~ T_1(x_1|x_0)      :704-705 */
    ais_log_p_d(c, s, x, R, 0.0, lp);                                        /* -log p_0(x_1)           :708 */
    for (int r = 0; r < R; ++r) lz[r] -= lp[r];
    double beta = db; uint32_t step = 1;
    while (beta < 1.0 - db + 1e-5) {                                      /* :710-726 */
        ais_log_p_d(c, s, x, R, beta, lp);
        for (int r = 0; r < R; ++r) lz[r] += lp[r];
        AIS_TRANSIT(beta + db, step)
        ++step;
        ais_log_p_d(c, s, x, R, beta, lp);
        for (int r = 0; r < R; ++r) lz[r] -= lp[r];
        beta = beta + db;
    }
    ais_log_p_d(c, s, x, R, 1.0, lp);                                        /* + log p_M(x_M)          :728 */
    const double logZ0 = (double)(V + H1 + H2) * (double)0.693147182464599609375f;   /* :731-734: tf.log(2.) is a float32 node, cast to the model dtype afterwards */
    for (int r = 0; r < R; ++r) values[r] = (double)(lz[r] + lp[r] + logZ0);
    free(x); free(xn); free(v); free(h2); free(lz); free(lp); free(Wt0); free(Wt1);
#undef AIS_TRANSIT
}

/* variational lower bound terms per row (dbm.py:738-759) given mu from mean-field, double accumulation */
void orc_dbm_log_proba_d(const orc_dbm_cfg_d *c, orc_dbm_state_d *s, const double *X, double *out) {
    orc_dbm_mean_field_d(c, s, X);
    const int V = c->V, H1 = c->n[0], H2 = c->n[1], N = c->N;
    for (int r = 0; r < N; ++r) {
        const double *x = X + (size_t)r * V, *m0 = s->mu[0] + (size_t)r * H1, *m1 = s->mu[1] + (size_t)r * H2;
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
        out[r] = (double)(e + ent);
    }
}

