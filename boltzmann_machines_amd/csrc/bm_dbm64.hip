// bm_dbm64.hip — the DBM path in float64 (bm_dbm64_* entry points of include/bm355.h).
//
// The reference's dtype is a constructor argument (base/mixin.py:14-25) and the DBM graph is built "all in model dtype"
// (dbm.py:294-383): DBM(dtype='float64') computes every tensor, every hyper-parameter and every random draw in double.
// This is the path for that dtype: the same graph as csrc/bm_dbm.hip (dbm.py:385-427 sweep, :429-478 mean-field, :480-509
// particles, :511-639 train op with the sparsity quirk and max-norm, :641-648 sample_v, :650-736 AIS, :738-759 ELBO), on
// the FP64 tile engine of csrc/bm_rbm64.hip (v_mfma_f64_16x16x4_f64; every dot product the SEQUENTIAL ascending-k fma
// chain) - bit-identical to oracle/bm_oracle_dbm64.c for everything that feeds back into state.  A compatibility path:
// dense matrices, one launch per pass, the mean-field loop driven by the host (one 8-byte read per sweep); Bernoulli
// hidden layers, Bernoulli or Gaussian visible units; one process (no exchange).
#include "../../include/bm355.h"
#include "bm_common.h"
#include "bm_rng.h"

#include <math.h>
#include <string>
#include <vector>

namespace bm64 {

// one layer update: out = act(mult * (Q1.P1 [+ Q2.P2]) + bmult * bias), P k-major [K][I], Q rows [J][K]
struct DActArgs {
    const double *P1; int ldp1; const double *Q1; int ldq1; int K1;
    const double *P2; int ldp2; const double *Q2; int ldq2; int K2;      // K2 == 0: absent
    int I, J;
    const double *bias, *sigma;
    double mult, bmult;
    int kind, sample;                 // kind: BM_UNIT_BERNOULLI | BM_UNIT_GAUSSIAN | 2 = raw mult*z + bmult*b
    double *means, *states;           // dense [J][I], may be null
    PhiloxKey key; long long row0;
    const double *prev;               // mean-field: previous mu (dense [J][I]) or null
    unsigned long long *maxdiff;      // atomicMax target for max |m - prev| (bits of a non-negative double)
};

__global__ __launch_bounds__(256, 1) void dact_kernel(DActArgs a) {
    __shared__ __attribute__((aligned(16))) double smem[2 * (T64_PBUF + T64_QBUF)];
    __shared__ double s_max[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wi = w & 1, wj = w >> 1, l15 = lane & 15, g = lane >> 4;
    const int i0 = blockIdx.x * T64_TI, j0 = blockIdx.y * T64_TJ;
    d4 acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    T64Seg sg[2] = {{a.P1, a.ldp1, a.Q1, a.ldq1, a.K1, 1.0}, {a.P2, a.ldp2, a.Q2, a.ldq2, a.K2, 1.0}};
    t64_mainloop<false>(acc, sg, a.K2 > 0 ? 2 : 1, a.I, a.J, i0, j0, smem);
    const int j = j0 + wj * 16 + l15;
    double dmax = 0.0;
    if (j < a.J) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wi * 32 + 16 * t + 4 * r + g;
                if (i >= a.I) continue;
                const double b = a.bmult * a.bias[i];
                const double x = a.mult * acc[t][r];
                const double m = (a.kind == BM_UNIT_BERNOULLI) ? sigmoid(x + b) : (a.kind == 2 ? x + b : (x * a.sigma[i] + b));
                double s = m;
                if (a.sample) {
                    const unsigned long long idx = (unsigned long long)(a.row0 + j) * (unsigned long long)a.I + (unsigned long long)i;
                    if (a.kind == BM_UNIT_BERNOULLI) s = (uniform_at(a.key, idx) < m) ? 1.0 : 0.0;
                    else s = normal_at(a.key, idx) * a.sigma[i] + m;
                }
                const size_t e = (size_t)j * a.I + i;
                if (a.prev) dmax = fmax(dmax, fabs(m - a.prev[e]));
                if (a.means) a.means[e] = m;
                if (a.states) a.states[e] = s;
            }
    }
    if (a.maxdiff) {                                   // block-uniform: one atomic per workgroup
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, off));
        if (lane == 0) s_max[w] = dmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double m = fmax(fmax(s_max[0], s_max[1]), fmax(s_max[2], s_max[3]));
            if (m > 0.0) atomicMax(a.maxdiff, (unsigned long long)__double_as_longlong(m));
        }
    }
}

// max |A - B| over a dense matrix -> atomicMax (the step-0 condition of the mean-field loop, dbm.py:449-452)
__global__ __launch_bounds__(256) void dmaxabsdiff_kernel(const double *A, const double *B, size_t n, unsigned long long *out) {
    __shared__ double s_max[4];
    double m = 0.0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) m = fmax(m, fabs(A[e] - B[e]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmax(fmax(s_max[0], s_max[1]), fmax(s_max[2], s_max[3]));
        if (m > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
    }
}

// raw outer product out[j][i] = sum_b Q[b][j] * P[b][i], rows b ascending (one fma chain per output)
struct DOuterArgs { const double *P; int I; const double *Q; int J; int B; double *out; };
__global__ __launch_bounds__(256, 1) void douter_kernel(DOuterArgs a) {
    __shared__ __attribute__((aligned(16))) double smem[2 * (T64_PBUF + T64_QBUF)];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wi = w & 1, wj = w >> 1, l15 = lane & 15, g = lane >> 4;
    const int i0 = blockIdx.x * T64_TI, j0 = blockIdx.y * T64_TJ;
    d4 acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    T64Seg sg = {a.P, a.I, a.Q, a.J, a.B, 1.0};
    t64_mainloop<true>(acc, &sg, 1, a.I, a.J, i0, j0, smem);
    const int j = j0 + wj * 16 + l15;
    if (j >= a.J) return;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + wi * 32 + 16 * t + 4 * r + g;
            if (i < a.I) a.out[(size_t)j * a.I + i] = acc[t][r];
        }
}

// column sums, sequential over the rows (oracle: colsum_diff_d): one thread per column
__global__ void dcolsum_kernel(const double *A, int rows, int cols, double *out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    double s = 0.0;
    for (int b = 0; b < rows; ++b) s = s + A[(size_t)b * cols + c];
    out[c] = s;
}

// visible bias (dbm.py:553, :597-600)
__global__ void dvbias_kernel(const double *sx, const double *sv, double *vb, double *dvb, int V, double N, double M, double lr, double mom) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const double g = sx[v] / N - sv[v] / M;
    const double d = lr * (mom * dvb[v] + g);
    dvb[v] = d;
    vb[v] = vb[v] + d;
}
// hidden bias + running means + penalties of layer `layer` (dbm.py:573-590, incl. the q_means[i] scalar-index quirk:
// the column sums are read at index `layer`, not at the unit's own index)
__global__ void dhbias_kernel(const double *smu, const double *sH, double *hb, double *dhb, double *q, double *mm, double *pen,
                              int n, int layer, double N, double M, double lr, double mom, double damping, double cost, double target) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n) return;
    double g = smu[h] / N - sH[h] / M;
    const double qn = damping * q[h] + (1.0 - damping) * sH[layer];
    const double mn = damping * mm[h] + (1.0 - damping) * smu[layer];
    q[h] = qn;
    mm[h] = mn;
    const double p1 = cost * (qn - target);
    const double p2 = cost * (mn - target);
    const double p = p1 + p2;
    pen[h] = p;
    g = g - p;
    const double d = lr * (mom * dhb[h] + g);
    dhb[h] = d;
    hb[h] = hb[h] + d;
}
// weights (dbm.py:556-558, :590, :604-605)
__global__ void dwupdate_kernel(const double *pos, const double *neg, const double *pen, double *W, double *dW, int J, int I,
                                double N, double M, double l2, double lr, double mom) {
    const size_t n = (size_t)J * I;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int h = (int)(e % (size_t)I);
        double g = pos[e] / N - neg[e] / M;
        g = g - l2 * W[e];
        g = g - pen[h];
        const double d = lr * (mom * dW[e] + g);
        dW[e] = d;
        W[e] = W[e] + d;
    }
}
// max-norm (dbm.py:511-513, :606-607): column norm as one fma chain over the rows, then the rescale; one thread per column
__global__ void dmaxnorm_kernel(double *W, int J, int I, double max_norm, double *wnorm) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= I) return;
    double acc = 0.0;
    for (int j = 0; j < J; ++j) { const double w = W[(size_t)j * I + h]; acc = fma(w, w, acc); }
    const double nrm = sqrt(acc);
    const double num = fmin(nrm, max_norm), den = fmax(nrm, 1e-8);
    for (int j = 0; j < J; ++j) { const size_t e = (size_t)j * I + h; W[e] = (W[e] * num) / den; }
    wnorm[h] = nrm;
}

// ---- AIS log p*_beta(x) per chain (dbm.py:650-660): one workgroup per chain, fixed-order tree sums
__device__ __forceinline__ double softplus64(double x) { return fmax(x, 0.0) + log1p(exp(-fabs(x))); }
__device__ __forceinline__ double block_sum256(double v, double *s) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s[0] + s[1]) + (s[2] + s[3]);
}
__global__ __launch_bounds__(256) void dais_logp_kernel(const double *x, int R, int V, int H1, int H2, const double *W0, const double *W1,
                                                        const double *vb, const double *hb0, const double *hb1, double beta,
                                                        double *logw, double sign) {
    __shared__ double s[4];
    extern __shared__ double xs[];                  // this chain's x [H1]
    const int r = blockIdx.x;
    for (int h = threadIdx.x; h < H1; h += 256) xs[h] = x[(size_t)r * H1 + h];
    __syncthreads();
    double t1 = 0.0, lp = 0.0;
    for (int h = threadIdx.x; h < H1; h += 256) t1 += xs[h] * hb0[h];
    for (int v = threadIdx.x; v < V; v += 256) {
        double z = vb[v];
        for (int h = 0; h < H1; ++h) z += xs[h] * W0[(size_t)v * H1 + h];
        lp += softplus64(z * beta);
    }
    for (int k2 = threadIdx.x; k2 < H2; k2 += 256) {
        double z = hb1[k2];
        for (int h = 0; h < H1; ++h) z += xs[h] * W1[(size_t)h * H2 + k2];
        lp += softplus64(z * beta);
    }
    const double T1 = block_sum256(t1, s), LP = block_sum256(lp, s);
    if (threadIdx.x == 0) logw[r] += sign * (T1 * beta + LP);
}
__global__ void dais_x0_kernel(double *x, int R, int H1, PhiloxKey key, long long chain0) {
    const size_t n = (size_t)R * H1;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long idx = (unsigned long long)chain0 * (unsigned long long)H1 + e;
        x[e] = (uniform_at(key, idx) < 0.5) ? 1.0 : 0.0;
    }
}
// ELBO terms per row (dbm.py:738-759): one workgroup per row
__global__ __launch_bounds__(256) void delbo_kernel(const double *X, int V, const double *mu0, int H1, const double *mu1, int H2,
                                                    const double *W0, const double *W1, const double *vb, const double *hb0,
                                                    const double *hb1, double *out) {
    __shared__ double s[4];
    const int r = blockIdx.x;
    const double *x = X + (size_t)r * V, *m0 = mu0 + (size_t)r * H1, *m1 = mu1 + (size_t)r * H2;
    double e = 0.0;
    for (int h = threadIdx.x; h < H1; h += 256) {
        double z = 0.0;
        for (int v = 0; v < V; ++v) z += x[v] * W0[(size_t)v * H1 + h];
        e += z * m0[h];
        e += m0[h] * hb0[h];
        const double q = fmin(fmax(m0[h], 1e-7), 1.0 - 1e-7);
        e += -q * log(q) - (1.0 - q) * log(1.0 - q);
    }
    for (int k2 = threadIdx.x; k2 < H2; k2 += 256) {
        double z = 0.0;
        for (int h = 0; h < H1; ++h) z += m0[h] * W1[(size_t)h * H2 + k2];
        e += z * m1[k2];
        e += m1[k2] * hb1[k2];
        const double q = fmin(fmax(m1[k2], 1e-7), 1.0 - 1e-7);
        e += -q * log(q) - (1.0 - q) * log(1.0 - q);
    }
    for (int v = threadIdx.x; v < V; v += 256) e += x[v] * vb[v];
    const double E = block_sum256(e, s);
    if (threadIdx.x == 0) out[r] = E;
}

}  // namespace bm64

constexpr int MAXL64 = BM_DBM_MAX_LAYERS;
struct bm_dbm64 {
    bm_dbm_config cfg;
    double mf_tol, l2, max_norm, damping, sp_target[MAXL64], sp_cost[MAXL64];
    int L, V, N, M, n[MAXL64 + 1];
    hipStream_t stream = nullptr;
    bm64::DBuf W[MAXL64], Wt[MAXL64], dW[MAXL64], hb[MAXL64], dhb[MAXL64], q[MAXL64], mm[MAXL64], pen[MAXL64], wnorm[MAXL64];
    bm64::DBuf vb, dvb, sigma;
    bm64::DBuf mu[MAXL64], mu_alt[MAXL64], mu_new[MAXL64], H[MAXL64], H_new[MAXL64], v, v_new, recon;
    bm64::DBuf pos[MAXL64], neg[MAXL64], sums;            // raw outer products; column sums [2V + 2 sum n_i]
    bm64::DBuf ax, ax2, av, ah2, alogw; int ais_rows = 0;
    unsigned long long *flag = nullptr; double *scal = nullptr;
    uint64_t seed = 0; uint32_t call = 0; int64_t prow0 = 0;
    bool wt_valid = false;
};

namespace bm64 {
static PhiloxKey dkey64(uint32_t site, int t, uint64_t seed, uint32_t call) {
    PhiloxKey k;
    k.k0 = (uint32_t)seed; k.k1 = (uint32_t)(seed >> 32);
    k.site = site + 16u * (uint32_t)t;
    k.call = call;
    return k;
}
enum : uint32_t { S_DBM_H = 8, S_DBM_V = 12, S_AIS_X0 = 13 };

static void ensure_wt(bm_dbm64 *h) {
    if (h->wt_valid) return;
    for (int i = 0; i < h->L; ++i) {
        const int a = h->n[i], b = h->n[i + 1];
        hipLaunchKernelGGL(transpose_kernel, dim3((a * b + 255) / 256), dim3(256), 0, h->stream, (const double *)h->W[i].p, h->Wt[i].p, a, b);
    }
    h->wt_valid = true;
}

struct In64 { const double *p; };
// layer >= 0: hidden layer `layer` from below (x W_layer) and, optionally, above (x W_{layer+1}^T); -1: visible from h0
static void layer_update(bm_dbm64 *h, int layer, int J, const double *below, const double *above, double mult, double bmult,
                         int sample, double *means, double *states, const PhiloxKey &key, int64_t row0,
                         const double *prev = nullptr, unsigned long long *maxdiff = nullptr, int kind_override = -1) {
    DActArgs a;
    memset(&a, 0, sizeof(a));
    if (layer >= 0) {
        a.I = h->n[layer + 1];
        a.P1 = h->W[layer].p; a.ldp1 = a.I; a.Q1 = below; a.K1 = h->n[layer]; a.ldq1 = a.K1;
        if (above) { a.P2 = h->Wt[layer + 1].p; a.ldp2 = a.I; a.Q2 = above; a.K2 = h->n[layer + 2]; a.ldq2 = a.K2; }
        a.bias = h->hb[layer].p; a.sigma = nullptr; a.kind = BM_UNIT_BERNOULLI;
    } else {
        a.I = h->V;
        a.P1 = h->Wt[0].p; a.ldp1 = a.I; a.Q1 = above; a.K1 = h->n[1]; a.ldq1 = a.K1;
        a.bias = h->vb.p; a.sigma = h->sigma.p; a.kind = h->cfg.v_unit;
    }
    if (kind_override >= 0) a.kind = kind_override;
    a.J = J; a.mult = mult; a.bmult = bmult; a.sample = sample;
    a.means = means; a.states = states; a.key = key; a.row0 = row0; a.prev = prev; a.maxdiff = maxdiff;
    hipLaunchKernelGGL(dact_kernel, dim3((a.I + T64_TI - 1) / T64_TI, (J + T64_TJ - 1) / T64_TJ), dim3(256), 0, h->stream, a);
}

// `_make_gibbs_step` (dbm.py:385-427): NEW below / OLD above
static void gibbs_sweep(bm_dbm64 *h, int J, const double *vin, DBuf *Hin, double *vout, DBuf *Hout, bool update_v, bool sample,
                        int t, int64_t row0, unsigned long long *maxdiff = nullptr) {
    for (int i = 0; i < h->L; ++i) {
        const double *below = (i == 0) ? vin : Hout[i - 1].p;
        const double *above = (i + 1 < h->L) ? Hin[i + 1].p : nullptr;
        const int smp = sample && h->cfg.sample_h_states[i];
        layer_update(h, i, J, below, above, 1.0, 1.0, smp, smp ? nullptr : Hout[i].p, smp ? Hout[i].p : nullptr,
                     dkey64(S_DBM_H + i, t, h->seed, h->call), row0, maxdiff ? Hin[i].p : nullptr, maxdiff);
    }
    if (update_v) {
        const int smp = sample && h->cfg.sample_v_states;
        layer_update(h, -1, J, nullptr, Hout[0].p, 1.0, 1.0, smp, smp ? nullptr : vout, smp ? vout : nullptr,
                     dkey64(S_DBM_V, t, h->seed, h->call), row0);
    }
}

static int read_flag(bm_dbm64 *h, double *out) {
    unsigned long long bits = 0;
    BM_HIP(hipMemcpyAsync(&bits, h->flag, sizeof(bits), hipMemcpyDeviceToHost, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
    memcpy(out, &bits, sizeof(double));
    return 0;
}

// `_make_mf` (dbm.py:429-478): result in h->mu, returns the executed sweeps
static int mean_field(bm_dbm64 *h, const double *X_dev, int *out_n) {
    const int L = h->L, N = h->N;
    ensure_wt(h);
    for (int i = 0; i < L; ++i) {                               // approximate-inference init into the mu_new variables
        const double *below = (i == 0) ? X_dev : h->mu_new[i - 1].p;
        const double mult = (i == 0 || i < L - 1) ? 2.0 : 1.0;
        layer_update(h, i, N, below, nullptr, mult, 1.0, 0, h->mu_new[i].p, nullptr, dkey64(0, 0, h->seed, h->call), 0);
    }
    BM_HIP(hipMemsetAsync(h->flag, 0, sizeof(unsigned long long), h->stream));
    for (int i = 0; i < L; ++i) {
        const size_t n = (size_t)N * h->n[i + 1];
        hipLaunchKernelGGL(dmaxabsdiff_kernel, dim3(64), dim3(256), 0, h->stream, (const double *)h->mu[i].p, (const double *)h->mu_new[i].p, n, h->flag);
    }
    double diff = 0.0;
    BM_TRY(read_flag(h, &diff));
    DBuf *cur = h->mu, *alt = h->mu_alt;
    int step = 0;
    while (step < h->cfg.max_mf_updates && diff > h->mf_tol) {
        BM_HIP(hipMemsetAsync(h->flag, 0, sizeof(unsigned long long), h->stream));
        gibbs_sweep(h, N, X_dev, cur, nullptr, alt, false, false, 0, 0, h->flag);
        BM_TRY(read_flag(h, &diff));
        DBuf *t = cur; cur = alt; alt = t;
        ++step;
    }
    if (cur != h->mu) for (int i = 0; i < L; ++i) { DBuf t = h->mu[i]; h->mu[i] = h->mu_alt[i]; h->mu_alt[i] = t; }
    BM_HIP(hipGetLastError());
    if (out_n) *out_n = step;
    return 0;
}

static void particles_update(bm_dbm64 *h, int k, bool sample, int t0 = 0) {
    ensure_wt(h);
    for (int t = 0; t < k; ++t) {
        gibbs_sweep(h, h->M, h->v.p, h->H, h->v_new.p, h->H_new, true, sample, t0 + t, h->prow0);
        DBuf tv = h->v; h->v = h->v_new; h->v_new = tv;
        for (int i = 0; i < h->L; ++i) { DBuf th = h->H[i]; h->H[i] = h->H_new[i]; h->H_new[i] = th; }
    }
}

static void reconstruct_from_mu(bm_dbm64 *h, double *R) {
    ensure_wt(h);
    layer_update(h, -1, h->N, nullptr, h->mu[0].p, 1.0, 1.0, 0, R, nullptr, dkey64(0, 0, h->seed, h->call), 0);
}

// gradients + sparsity + momentum + max-norm (dbm.py:550-621); every raw sum is taken from the PRE-update state first
static int apply_update(bm_dbm64 *h, const double *X_dev, double lr, double mom) {
    const int L = h->L;
    const double N = (double)h->N, M = (double)h->M;
    double *sx = h->sums.p, *sv = sx + h->V, *tail = sv + h->V;
    hipLaunchKernelGGL(dcolsum_kernel, dim3((h->V + 255) / 256), dim3(256), 0, h->stream, X_dev, h->N, h->V, sx);
    hipLaunchKernelGGL(dcolsum_kernel, dim3((h->V + 255) / 256), dim3(256), 0, h->stream, (const double *)h->v.p, h->M, h->V, sv);
    for (int i = 0; i < L; ++i) {
        const int J = h->n[i], I = h->n[i + 1];
        DOuterArgs p = {h->mu[i].p, I, (i == 0) ? X_dev : h->mu[i - 1].p, J, h->N, h->pos[i].p};
        DOuterArgs q = {h->H[i].p, I, (i == 0) ? h->v.p : h->H[i - 1].p, J, h->M, h->neg[i].p};
        const dim3 grid((I + T64_TI - 1) / T64_TI, (J + T64_TJ - 1) / T64_TJ);
        hipLaunchKernelGGL(douter_kernel, grid, dim3(256), 0, h->stream, p);
        hipLaunchKernelGGL(douter_kernel, grid, dim3(256), 0, h->stream, q);
    }
    hipLaunchKernelGGL(dvbias_kernel, dim3((h->V + 255) / 256), dim3(256), 0, h->stream, (const double *)sx, (const double *)sv,
                       h->vb.p, h->dvb.p, h->V, N, M, lr, mom);
    for (int i = 0; i < L; ++i) {
        const int n = h->n[i + 1];
        double *smu = tail, *sH = tail + n;
        tail += 2 * n;
        hipLaunchKernelGGL(dcolsum_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, (const double *)h->mu[i].p, h->N, n, smu);
        hipLaunchKernelGGL(dcolsum_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, (const double *)h->H[i].p, h->M, n, sH);
        hipLaunchKernelGGL(dhbias_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, (const double *)smu, (const double *)sH,
                           h->hb[i].p, h->dhb[i].p, h->q[i].p, h->mm[i].p, h->pen[i].p, n, i, N, M, lr, mom,
                           h->damping, h->sp_cost[i], h->sp_target[i]);
    }
    for (int i = 0; i < L; ++i) {
        const int J = h->n[i], I = h->n[i + 1];
        hipLaunchKernelGGL(dwupdate_kernel, dim3(512), dim3(256), 0, h->stream, (const double *)h->pos[i].p, (const double *)h->neg[i].p,
                           (const double *)h->pen[i].p, h->W[i].p, h->dW[i].p, J, I, N, M, h->l2, lr, mom);
        hipLaunchKernelGGL(dmaxnorm_kernel, dim3((I + 63) / 64), dim3(64), 0, h->stream, h->W[i].p, J, I, h->max_norm, h->wnorm[i].p);
    }
    h->wt_valid = false;
    BM_HIP(hipGetLastError());
    return 0;
}

static int msre_of_mu(bm_dbm64 *h, const double *X_dev, double *out) {
    reconstruct_from_mu(h, h->recon.p);
    const size_t n = (size_t)h->N * h->V;
    hipLaunchKernelGGL(sqdiff_kernel, dim3(64), dim3(256), 0, h->stream, X_dev, (const double *)h->recon.p, n, h->scal + 8);
    ReduceJobs jb;
    memset(&jb, 0, sizeof(jb));
    jb.off[0] = 8; jb.cnt[0] = 64;
    hipLaunchKernelGGL(reduce_fixed_kernel, dim3(1), dim3(256), 0, h->stream, (const double *)h->scal, jb, h->scal);
    double se = 0.0;
    BM_HIP(hipMemcpyAsync(&se, h->scal, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
    *out = se / ((double)h->N * h->V);
    return 0;
}

static DBuf *find(bm_dbm64 *h, const std::string &name, size_t *n) {
    std::string base = name; int idx = 0;
    const size_t us = name.rfind('_');
    if (us != std::string::npos && us + 1 < name.size() && isdigit((unsigned char)name[us + 1]) && name.find_first_not_of("0123456789", us + 1) == std::string::npos) {
        base = name.substr(0, us); idx = atoi(name.c_str() + us + 1);
    }
    if (idx < 0 || idx >= h->L) return nullptr;
    const int lo = h->n[idx], hi = h->n[idx + 1];
    if (base == "W") { *n = (size_t)lo * hi; return &h->W[idx]; }
    if (base == "dW") { *n = (size_t)lo * hi; return &h->dW[idx]; }
    if (base == "hb") { *n = hi; return &h->hb[idx]; }
    if (base == "dhb") { *n = hi; return &h->dhb[idx]; }
    if (base == "q_means") { *n = hi; return &h->q[idx]; }
    if (base == "mu_means") { *n = hi; return &h->mm[idx]; }
    if (base == "W_norm") { *n = hi; return &h->wnorm[idx]; }
    if (base == "mu") { *n = (size_t)h->N * hi; return &h->mu[idx]; }
    if (base == "mu_new") { *n = (size_t)h->N * hi; return &h->mu_new[idx]; }
    if (base == "h") { *n = (size_t)h->M * hi; return &h->H[idx]; }
    if (base == "h_new") { *n = (size_t)h->M * hi; return &h->H_new[idx]; }
    if (idx == 0 && name.find('_') == std::string::npos) {
        if (base == "vb") { *n = h->V; return &h->vb; }
        if (base == "dvb") { *n = h->V; return &h->dvb; }
        if (base == "sigma") { *n = h->V; return &h->sigma; }
        if (base == "v") { *n = (size_t)h->M * h->V; return &h->v; }
    }
    if (name == "v_new") { *n = (size_t)h->M * h->V; return &h->v_new; }
    return nullptr;
}
}  // namespace bm64

extern "C" {

// hyper12 = {mf_tol, l2, max_norm, sparsity_damping, sparsity_target[4], sparsity_cost[4]} as doubles; NULL: from cfg
int bm_dbm64_create(const bm_dbm_config *cfg, const double *hyper12, bm_dbm64 **out) {
    BM_CHECK(cfg && out, "null argument");
    int ndev = 0;
    BM_CHECK(hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0, "no HIP device visible: bm355 has no CPU fallback");
    BM_CHECK(cfg->n_layers >= 1 && cfg->n_layers <= MAXL64, "n_layers must be in [1, %d]", MAXL64);
    BM_CHECK(cfg->v_unit == BM_UNIT_BERNOULLI || cfg->v_unit == BM_UNIT_GAUSSIAN, "float64 DBM: Bernoulli or Gaussian visible units");
    for (int i = 0; i < cfg->n_layers; ++i)
        BM_CHECK(cfg->h_unit[i] == BM_UNIT_BERNOULLI, "float64 DBM: Bernoulli hidden layers only (Multinomial layers: the float32 path)");
    BM_CHECK(cfg->n_visible >= 1 && cfg->batch_size >= 1 && cfg->n_particles >= 1, "bad sizes");
    bm_dbm64 *h = new bm_dbm64();
    h->cfg = *cfg;
    h->L = cfg->n_layers; h->V = cfg->n_visible; h->N = cfg->batch_size; h->M = cfg->n_particles;
    h->n[0] = h->V;
    for (int i = 0; i < h->L; ++i) { BM_CHECK(cfg->n_hiddens[i] >= 1, "bad layer size"); h->n[i + 1] = cfg->n_hiddens[i]; }
    h->mf_tol = hyper12 ? hyper12[0] : (double)cfg->mf_tol;
    h->l2 = hyper12 ? hyper12[1] : (double)cfg->l2;
    h->max_norm = hyper12 ? hyper12[2] : (double)cfg->max_norm;
    h->damping = hyper12 ? hyper12[3] : (double)cfg->sparsity_damping;
    for (int i = 0; i < MAXL64; ++i) {
        h->sp_target[i] = hyper12 ? hyper12[4 + i] : (double)cfg->sparsity_target[i];
        h->sp_cost[i] = hyper12 ? hyper12[8 + i] : (double)cfg->sparsity_cost[i];
    }
    BM_HIP(hipStreamCreate(&h->stream));
    size_t nsum = 2 * (size_t)h->V;
    for (int i = 0; i < h->L; ++i) {
        const size_t a = h->n[i], b = h->n[i + 1];
        BM_TRY(h->W[i].alloc(a * b)); BM_TRY(h->Wt[i].alloc(a * b)); BM_TRY(h->dW[i].alloc(a * b));
        BM_TRY(h->pos[i].alloc(a * b)); BM_TRY(h->neg[i].alloc(a * b));
        for (bm64::DBuf *v : {&h->hb[i], &h->dhb[i], &h->q[i], &h->mm[i], &h->pen[i], &h->wnorm[i]}) BM_TRY(v->alloc(b));
        for (bm64::DBuf *v : {&h->mu[i], &h->mu_alt[i], &h->mu_new[i]}) BM_TRY(v->alloc((size_t)h->N * b));
        for (bm64::DBuf *v : {&h->H[i], &h->H_new[i]}) BM_TRY(v->alloc((size_t)h->M * b));
        nsum += 2 * b;
    }
    BM_TRY(h->vb.alloc(h->V)); BM_TRY(h->dvb.alloc(h->V)); BM_TRY(h->sigma.alloc(h->V));
    BM_TRY(h->v.alloc((size_t)h->M * h->V)); BM_TRY(h->v_new.alloc((size_t)h->M * h->V)); BM_TRY(h->recon.alloc((size_t)h->N * h->V));
    BM_TRY(h->sums.alloc(nsum));
    {   // sigma = 1
        std::vector<double> one((size_t)h->V, 1.0);
        BM_HIP(hipMemcpy(h->sigma.p, one.data(), one.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    BM_HIP(hipMalloc((void **)&h->flag, 64));
    BM_HIP(hipMalloc((void **)&h->scal, 128 * sizeof(double)));
    BM_HIP(hipMemset(h->scal, 0, 128 * sizeof(double)));
    *out = h;
    return 0;
}
int bm_dbm64_destroy(bm_dbm64 *h) {
    if (!h) return 0;
    (void)hipStreamSynchronize(h->stream);
    if (h->flag) (void)hipFree(h->flag);
    if (h->scal) (void)hipFree(h->scal);
    (void)hipStreamDestroy(h->stream);
    for (int i = 0; i < MAXL64; ++i)
        for (bm64::DBuf *b : {&h->W[i], &h->Wt[i], &h->dW[i], &h->hb[i], &h->dhb[i], &h->q[i], &h->mm[i], &h->pen[i], &h->wnorm[i],
                              &h->mu[i], &h->mu_alt[i], &h->mu_new[i], &h->H[i], &h->H_new[i], &h->pos[i], &h->neg[i]}) b->release();
    for (bm64::DBuf *b : {&h->vb, &h->dvb, &h->sigma, &h->v, &h->v_new, &h->recon, &h->sums, &h->ax, &h->ax2, &h->av, &h->ah2, &h->alogw}) b->release();
    delete h;
    return 0;
}
int bm_dbm64_sync(bm_dbm64 *h) { BM_CHECK(h, "null argument"); BM_HIP(hipStreamSynchronize(h->stream)); return 0; }
int bm_dbm64_seed(bm_dbm64 *h, uint64_t seed) { BM_CHECK(h, "null argument"); h->seed = seed; h->call = 0; return 0; }
int bm_dbm64_set_row_offset(bm_dbm64 *h, int64_t row0, int64_t particle0) { BM_CHECK(h, "null argument"); (void)row0; h->prow0 = particle0; return 0; }

int bm_dbm64_set_param(bm_dbm64 *h, const char *name, const double *host, size_t n) {
    BM_CHECK(h && name && host, "null argument");
    size_t want = 0;
    bm64::DBuf *b = bm64::find(h, name, &want);
    BM_CHECK(b, "unknown variable '%s'", name);
    BM_CHECK(n == want, "variable '%s' has %zu elements, %zu given", name, want, n);
    BM_HIP(hipStreamSynchronize(h->stream));
    BM_HIP(hipMemcpy(b->p, host, n * sizeof(double), hipMemcpyHostToDevice));
    if (name[0] == 'W') h->wt_valid = false;
    return 0;
}
int bm_dbm64_get_param(bm_dbm64 *h, const char *name, double *host, size_t n) {
    BM_CHECK(h && name && host, "null argument");
    size_t want = 0;
    bm64::DBuf *b = bm64::find(h, name, &want);
    BM_CHECK(b, "unknown variable '%s'", name);
    BM_CHECK(n == want, "variable '%s' has %zu elements, %zu given", name, want, n);
    BM_HIP(hipStreamSynchronize(h->stream));
    BM_HIP(hipMemcpy(host, b->p, n * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

// session.run(train_op) (dbm.py:805): mean-field, k particle sweeps, [msre], update.  X_dev: [batch_size][n_visible] doubles
int bm_dbm64_train_step(bm_dbm64 *h, const double *X_dev, double lr, double mom, int32_t k, int32_t *out_n_mf, double *out_msre) {
    BM_CHECK(h && X_dev, "null argument");
    BM_CHECK(k >= 0, "n_gibbs_steps must be >= 0");
    int n = 0;
    BM_TRY(bm64::mean_field(h, X_dev, &n));
    bm64::particles_update(h, k, true);
    if (out_msre) BM_TRY(bm64::msre_of_mu(h, X_dev, out_msre));
    BM_TRY(bm64::apply_update(h, X_dev, lr, mom));
    h->call += 1;
    if (out_n_mf) *out_n_mf = n;
    return 0;
}
// validation fetch (dbm.py:813 under :521-523): mean-field, k particle sweeps, msre; no update
int bm_dbm64_metrics(bm_dbm64 *h, const double *X_dev, int32_t k, int32_t *out_n_mf, double *out_msre) {
    BM_CHECK(h && X_dev && out_msre, "null argument");
    int n = 0;
    BM_TRY(bm64::mean_field(h, X_dev, &n));
    bm64::particles_update(h, k, true);
    BM_TRY(bm64::msre_of_mu(h, X_dev, out_msre));
    h->call += 1;
    if (out_n_mf) *out_n_mf = n;
    return 0;
}
// transform (dbm.py:859-872): mean-field, the top layer's mu to out_dev [batch_size][n_top] (may be null)
int bm_dbm64_mean_field(bm_dbm64 *h, const double *X_dev, double *out_dev, int32_t *out_n_mf) {
    BM_CHECK(h && X_dev, "null argument");
    int n = 0;
    BM_TRY(bm64::mean_field(h, X_dev, &n));
    if (out_dev) BM_HIP(hipMemcpyAsync(out_dev, h->mu[h->L - 1].p, (size_t)h->N * h->n[h->L] * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    h->call += 1;
    if (out_n_mf) *out_n_mf = n;
    return 0;
}
int bm_dbm64_reconstruct(bm_dbm64 *h, const double *X_dev, double *R_dev) {
    BM_CHECK(h && X_dev && R_dev, "null argument");
    BM_TRY(bm64::mean_field(h, X_dev, nullptr));
    bm64::reconstruct_from_mu(h, R_dev);
    BM_HIP(hipGetLastError());
    h->call += 1;
    return 0;
}
// sample_v (dbm.py:641-648): k sampled sweeps on the particles (assigned), then k mean sweeps whose v is assigned
int bm_dbm64_sample_v(bm_dbm64 *h, int32_t k, double *V_dev) {
    BM_CHECK(h && k >= 0, "bad argument");
    bm64::particles_update(h, k, true);
    if (k > 0) {
        // the mean sweeps run on COPIES of the hidden particles; only v is assigned
        bm64::DBuf Hc[MAXL64], Hn[MAXL64], vc, vn;
        for (int i = 0; i < h->L; ++i) {
            const size_t n = (size_t)h->M * h->n[i + 1];
            BM_TRY(Hc[i].alloc(n)); BM_TRY(Hn[i].alloc(n));
            BM_HIP(hipMemcpyAsync(Hc[i].p, h->H[i].p, n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        }
        const size_t nv = (size_t)h->M * h->V;
        BM_TRY(vc.alloc(nv)); BM_TRY(vn.alloc(nv));
        BM_HIP(hipMemcpyAsync(vc.p, h->v.p, nv * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        bm64::DBuf *Hin = Hc, *Hout = Hn;
        double *vin = vc.p, *vout = vn.p;
        for (int t = 0; t < k; ++t) {
            bm64::gibbs_sweep(h, h->M, vin, Hin, vout, Hout, true, false, k + t, h->prow0);
            double *tv = vin; vin = vout; vout = tv;
            bm64::DBuf *th = Hin; Hin = Hout; Hout = th;
        }
        BM_HIP(hipMemcpyAsync(h->v.p, vin, nv * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        BM_HIP(hipStreamSynchronize(h->stream));        // the temporaries die here
        for (int i = 0; i < h->L; ++i) { Hc[i].release(); Hn[i].release(); }
        vc.release(); vn.release();
    }
    if (V_dev) BM_HIP(hipMemcpyAsync(V_dev, h->v.p, (size_t)h->M * h->V * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    BM_HIP(hipGetLastError());
    h->call += 1;
    return 0;
}

// AIS (dbm.py:696-736) for the 2-layer Bernoulli DBM; values_host [n_runs]: the per-chain log Z estimates (chains
// chain0 .. chain0 + n_runs - 1 of the RNG stream)
int bm_dbm64_ais(bm_dbm64 *h, int32_t n_betas, int32_t R, int32_t k, uint64_t seed, int64_t chain0, double *values_host) {
    BM_CHECK(h && values_host, "null argument");
    BM_CHECK(h->L == 2 && h->cfg.v_unit == BM_UNIT_BERNOULLI, "AIS needs a 2-layer Bernoulli DBM (dbm.py:925-927)");
    BM_CHECK(n_betas >= 1 && R >= 1 && k >= 1, "bad AIS arguments");
    const int V = h->V, H1 = h->n[1], H2 = h->n[2];
    if (h->ais_rows < R) {
        for (bm64::DBuf *b : {&h->ax, &h->ax2, &h->av, &h->ah2, &h->alogw}) b->release();
        BM_TRY(h->ax.alloc((size_t)R * H1)); BM_TRY(h->ax2.alloc((size_t)R * H1)); BM_TRY(h->av.alloc((size_t)R * V));
        BM_TRY(h->ah2.alloc((size_t)R * H2)); BM_TRY(h->alogw.alloc((size_t)R));
        h->ais_rows = R;
    }
    bm64::ensure_wt(h);
    const uint64_t seed0 = h->seed; const uint32_t call0 = h->call;
    h->seed = seed;                                       // the streams of an AIS run are keyed by ITS seed (oracle: orc_dbm_ais_d)
    hipLaunchKernelGGL(bm64::dais_x0_kernel, dim3(256), dim3(256), 0, h->stream, h->ax.p, R, H1, bm64::dkey64(bm64::S_AIS_X0, 0, seed, 0), (long long)chain0);
    BM_HIP(hipMemsetAsync(h->alogw.p, 0, (size_t)R * sizeof(double), h->stream));
    bm64::DBuf *x = &h->ax, *xn = &h->ax2;
    auto transit = [&](double beta, uint32_t step) {
        h->call = step;
        for (int t = 0; t < k; ++t) {
            bm64::layer_update(h, -1, R, nullptr, x->p, beta, beta, h->cfg.sample_v_states, nullptr, h->av.p,
                               bm64::dkey64(bm64::S_DBM_V, t, seed, step), chain0, nullptr, nullptr, BM_UNIT_BERNOULLI);
            bm64::layer_update(h, 1, R, x->p, nullptr, beta, beta, h->cfg.sample_h_states[1], nullptr, h->ah2.p,
                               bm64::dkey64(bm64::S_DBM_H + 1, t, seed, step), chain0);
            bm64::layer_update(h, 0, R, h->av.p, h->ah2.p, beta, beta, h->cfg.sample_h_states[0], nullptr, xn->p,
                               bm64::dkey64(bm64::S_DBM_H + 0, t, seed, step), chain0);
            bm64::DBuf *tx = x; x = xn; xn = tx;
        }
    };
    auto logp = [&](double beta, double sign) {
        hipLaunchKernelGGL(bm64::dais_logp_kernel, dim3(R), dim3(256), (size_t)H1 * sizeof(double), h->stream, (const double *)x->p, R, V, H1, H2,
                           (const double *)h->W[0].p, (const double *)h->W[1].p, (const double *)h->vb.p, (const double *)h->hb[0].p,
                           (const double *)h->hb[1].p, beta, h->alogw.p, sign);
    };
    const double db = 1.0 / (double)n_betas;
    transit(db, 0u);
    logp(0.0, -1.0);
    double beta = db; uint32_t step = 1;
    while (beta < 1.0 - db + 1e-5) {
        logp(beta, 1.0);
        transit(beta + db, step);
        ++step;
        logp(beta, -1.0);
        beta = beta + db;
    }
    logp(1.0, 1.0);
    h->seed = seed0; h->call = call0;
    BM_HIP(hipGetLastError());
    BM_HIP(hipMemcpyAsync(values_host, h->alogw.p, (size_t)R * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
    const double logZ0 = (double)(V + H1 + H2) * (double)0.693147182464599609375f;   // dbm.py:731-734: tf.log(2.) is a float32 node, cast afterwards
    for (int r = 0; r < R; ++r) values_host[r] += logZ0;
    return 0;
}
// ELBO terms per row (dbm.py:738-759) after a mean-field on X; out_host [batch_size]
int bm_dbm64_log_proba(bm_dbm64 *h, const double *X_dev, double *out_host) {
    BM_CHECK(h && X_dev && out_host, "null argument");
    BM_CHECK(h->L == 2, "log_proba needs a 2-layer DBM (dbm.py:947-948)");
    BM_TRY(bm64::mean_field(h, X_dev, nullptr));
    hipLaunchKernelGGL(bm64::delbo_kernel, dim3(h->N), dim3(256), 0, h->stream, X_dev, h->V, (const double *)h->mu[0].p, h->n[1],
                       (const double *)h->mu[1].p, h->n[2], (const double *)h->W[0].p, (const double *)h->W[1].p,
                       (const double *)h->vb.p, (const double *)h->hb[0].p, (const double *)h->hb[1].p, h->recon.p);
    BM_HIP(hipGetLastError());
    BM_HIP(hipMemcpyAsync(out_host, h->recon.p, (size_t)h->N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
    h->call += 1;
    return 0;
}

}  // extern "C"
