/* bm355.h — C-ABI of libbm355.so, the MI355X (gfx950) RBM/DBM engine.
 *
 * The reference (yell/boltzmann-machines) has no FFI: its only seam is
 * Python class API <-> `session.run`.  Every entry point
 * below replaces one `session.run` / `.eval` fetch site of the reference,
 * cited as  reference-file:line  next to the declaration.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; the message
 *     is available from bm_last_error() (thread-local).
 *   - `*_dev` pointers are DEVICE pointers (hipMalloc / bm_dev_alloc /
 *     torch .data_ptr()); everything else is host memory owned by the caller.
 *   - a handle owns all model state (W, biases, momentum buffers, running
 *     means, chain workspaces) in HBM and one HIP stream; calls on one handle
 *     are asynchronous on that stream and must not be issued concurrently
 *     from several host threads.  bm_*_sync() blocks until the stream drains.
 *   - all arithmetic is fp32 (reference default dtype, base/mixin.py:15).
 *   - matrices are row-major; W is [n_below, n_above] like the reference
 *     (base_rbm.py:277-293).
 */
#ifndef BM355_H
#define BM355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ misc */
const char *bm_last_error(void);
/* "bm355 <version> gfx950" */
const char *bm_version(void);
/* number of visible HIP devices (0 when no GPU / no driver) */
int bm_device_count(void);
int bm_set_device(int device);

/* raw device-memory helpers so that a host without torch can drive the ABI */
int bm_dev_alloc(size_t bytes, void **out_dev);
int bm_dev_free(void *dev);
int bm_h2d(void *dst_dev, const void *src_host, size_t bytes);
int bm_d2h(void *dst_host, const void *src_dev, size_t bytes);
int bm_dev_memset(void *dst_dev, int value, size_t bytes);

/* XCD-aware block -> tile map of the tile kernels (csrc/bm_gemm.h tile_of_block), evaluated on the host: for tests
 * and tools.  out_ti / out_tj: tile of every block of a tiles_i x tiles_j launch; out_map5 = {xi, xj, gj, tiles_i, tiles_j} */
int bm_debug_tile_map(int32_t tiles_i, int32_t tiles_j, double bytes_i, double bytes_j, int32_t *out_ti,
                      int32_t *out_tj, int32_t *out_map5);

/* ------------------------------------------------------------------- RBM */
typedef struct bm_rbm bm_rbm;

enum { BM_UNIT_BERNOULLI = 0, BM_UNIT_GAUSSIAN = 1, BM_UNIT_MULTINOMIAL = 2 };

/* ctor kwargs of BaseRBM.__init__ that influence the device graph
 * (rbm/base_rbm.py:95-105, :244-327). */
typedef struct bm_rbm_config {
    int32_t n_visible;
    int32_t n_hidden;
    int32_t v_unit;            /* BM_UNIT_*: BernoulliRBM / GaussianRBM (rbm/rbm.py:10-15,88-99) */
    int32_t sample_v_states;   /* base_rbm.py:100 */
    int32_t sample_h_states;   /* base_rbm.py:100 */
    int32_t dbm_first;         /* propup multiplier 1+dbm_first (base_rbm.py:256-260) */
    int32_t dbm_last;          /* propdown multiplier 1+dbm_last (base_rbm.py:261-262) */
    int32_t max_batch;         /* largest minibatch the workspaces must hold */
    float   l2;                /* base_rbm.py:249 */
    float   sparsity_target;   /* base_rbm.py:253-255 */
    float   sparsity_cost;
    float   sparsity_damping;
    float   dropout;           /* keep-prob of tf.nn.dropout; <0 => no dropout (base_rbm.py:417-418) */
    int32_t h_unit;            /* BM_UNIT_BERNOULLI, or BM_UNIT_MULTINOMIAL = MultinomialRBM (rbm/rbm.py:25-65,
                                  layers.py:54-70): means = n_samples*softmax, states = multinomial counts */
    int32_t n_samples;         /* MultinomialLayer.n_samples (rbm.py:46); ignored for Bernoulli hidden units */
} bm_rbm_config;

int bm_rbm_create(const bm_rbm_config *cfg, bm_rbm **out);
int bm_rbm_destroy(bm_rbm *h);
int bm_rbm_sync(bm_rbm *h);

/* Variables of the TF graph by name (tf_model.py:183-202 get_tf_params; Saver
 * restore tf_model.py:22-28): "W" [V*H], "vb" [V], "hb" [H], "dW", "dvb",
 * "dhb", "q_means" [H], "sigma" [V].  n = number of floats. */
int bm_rbm_set_param(bm_rbm *h, const char *name, const float *host, size_t n);
/* the same from DEVICE memory (dense, row-major), asynchronously on the handle's stream - no host synchronisation
 * (re-initialising between runs, `init_from` another handle: base_rbm.py:668-685 without a host round trip) */
int bm_rbm_set_param_dev(bm_rbm *h, const char *name, const float *src_dev, size_t n);
int bm_rbm_get_param(bm_rbm *h, const char *name, float *host, size_t n);
/* device pointer of a variable / workspace for zero-copy interop (RCCL):
 * the variables above plus "grad" (fused [V*H + V + H + H] raw-sum buffer). */
int bm_rbm_dev_ptr(bm_rbm *h, const char *name, void **out_dev, size_t *out_n);

/* Replaces tf.set_random_seed(model.make_random_seed()) done by the
 * run_in_tf_session decorator on every public call (tf_model.py:20-21).
 * Sets the Philox key and resets the per-handle call counter to 0.
 * Stream convention: DESIGN.md "RNG". */
int bm_rbm_seed(bm_rbm *h, uint64_t seed);
/* global index of local row 0 (data-parallel: rank*local_batch); sample
 * bitmaps depend on the GLOBAL row so they are rank-count invariant. */
int bm_rbm_set_row_offset(bm_rbm *h, int64_t row0);

/* One CD-k update: session.run(train_op, feed_dict) at base_rbm.py:566
 * (graph: base_rbm.py:415-479).  X_dev [B, n_visible] row-major. */
int bm_rbm_train_step(bm_rbm *h, const float *X_dev, int32_t B,
                      float learning_rate, float momentum, int32_t n_gibbs_steps);
/* Same update, but the metric ops are fetched from the SAME chain in the same
 * run, as the reference does every `train_metrics_every_iter` iterations
 * (base_rbm.py:554-564).  out4 as in bm_rbm_metrics (computed before the update). */
int bm_rbm_train_step_metrics(bm_rbm *h, const float *X_dev, int32_t B,
                              float learning_rate, float momentum, int32_t n_gibbs_steps,
                              float *out4);
/* The same fetch without the host wait.  The reference only uses the MEAN of the train metrics at the end
 * of the epoch (base_rbm.py:571), so the sums of a metrics iteration are copied to a pinned ring in stream
 * order; bm_rbm_collect_metrics synchronises once and returns the pending fetches in order
 * (out4n [max_n][4], *out_n their number; at most 4096 may be pending). */
int bm_rbm_train_step_metrics_async(bm_rbm *h, const float *X_dev, int32_t B,
                                    float learning_rate, float momentum, int32_t n_gibbs_steps);
int bm_rbm_collect_metrics(bm_rbm *h, float *out4n, int32_t max_n, int32_t *out_n);
/* The `for X_batch in batch_iter(X, batch_size)` loop of _train_epoch
 * (base_rbm.py:549-571) behind ONE call: N rows, consecutive batches of `batch` rows (last one may be
 * short).  The library enqueues every update's four launches from a native loop, asynchronously on the
 * handle's stream (no synchronisation, no device-to-host traffic, one FFI crossing per run of batches);
 * the stream stays GPU-bound. */
int bm_rbm_train_epoch(bm_rbm *h, const float *X_dev, int64_t N, int32_t batch,
                       float learning_rate, float momentum, int32_t n_gibbs_steps);

/* A snapshot of every variable that does not stop the stream (the per-epoch checkpoint of base_rbm.py:665-666 while
 * the next epoch already runs): bm_rbm_stage copies them device-to-device into slot 0 | 1 in stream order and returns;
 * bm_rbm_get_staged reads one variable of that snapshot (dense, like bm_rbm_get_param) on its own stream - it waits
 * for the staged copies only - and may be called from another host thread.  Do not re-stage a slot being read. */
int bm_rbm_stage(bm_rbm *h, int32_t slot);
int bm_rbm_get_staged(bm_rbm *h, int32_t slot, const char *name, float *host, size_t n);

/* Data-parallel split of a train step: phase 1 runs the chain and
 * leaves the raw un-normalised sums in the "grad" buffer
 * [pos-neg (V*H) | sum(X-v) (V) | sum(h0-hk) (H) | sum(hk) (H)]; the caller
 * all-reduces that buffer (RCCL) and calls phase 2 with the GLOBAL batch. */
int bm_rbm_grad_step(bm_rbm *h, const float *X_dev, int32_t B_local, int32_t n_gibbs_steps);
int bm_rbm_apply_step(bm_rbm *h, int32_t B_global, float learning_rate, float momentum);

/* transform_op.eval at base_rbm.py:697: h_means at the END of the k-step
 * chain (base_rbm.py:426,438-440).  H_dev [B, n_hidden]. */
int bm_rbm_transform(bm_rbm *h, const float *X_dev, int32_t B, int32_t n_gibbs_steps,
                     float *H_dev);

/* Metric fetches of base_rbm.py:554-564,578,605,612: out[0]=msre (:486-488),
 * out[1]=pll (:496-513), out[2]=l2_loss (:482-484), out[3]=free_energy
 * (:516-517; rbm.py:17-22 / :109-116).  Runs the same chain as train_step
 * without updating parameters.  `out` is host memory (4 floats). */
int bm_rbm_metrics(bm_rbm *h, const float *X_dev, int32_t B, int32_t n_gibbs_steps,
                   float *out4);
/* batch-mean free energy only (free_energy_op, base_rbm.py:516-517). */
int bm_rbm_free_energy(bm_rbm *h, const float *X_dev, int32_t B, float *out1);

/* Pure block-Gibbs sampling sweep (R5/R6 of SURVEY §8a; base_rbm.py:367-413):
 * n_steps of h->v->h starting from hidden states H_dev [B, n_hidden] (in/out);
 * V_dev [B, n_visible] receives the last visible states.  No parameter update. */
int bm_rbm_gibbs(bm_rbm *h, float *H_dev, float *V_dev, int32_t B, int32_t n_steps);

/* the handle's hipStream_t (as void*), so a host can enqueue collectives on
 * the same stream (torch.cuda.ExternalStream) without host synchronisation. */
int bm_rbm_stream(bm_rbm *h, void **out_stream);

/* Per-kernel-class HIP-event timing (bench.py roofline leg).  While enabled,
 * every launch is bracketed by events on the handle's stream;
 * bm_rbm_kernel_times() synchronises and returns, for class c in
 * {0: prop-up act_kernel, 1: prop-down act_kernel, 2: grad_kernel,
 *  3: colsum_kernel, 4: bias/apply kernels, 5: other}, the summed duration
 * ms[c] and the launch count n[c] since the last enable. */
#define BM_NUM_KERNEL_CLASSES 6
int bm_rbm_profile(bm_rbm *h, int32_t enable);
int bm_rbm_kernel_times(bm_rbm *h, float *ms6, int32_t *n6);

/* Chained launches (csrc/bm_chain.h): h0 and the k Gibbs steps of base_rbm.py:417-426 / the sweeps of bm_rbm_gibbs run as
 * ONE launch where the shape allows it (BM355_DEBUG=chain=0: never, 1: default rule, 2: wherever legal).  out3 = {chained
 * launches issued so far, tiles they must compute, mode}; bm_rbm_sync reports a launch that did not complete as an error. */
int bm_rbm_chain_stats(bm_rbm *h, int64_t *out3);

/* HIP-event timer on the handle's stream (bench.py roofline leg). */
int bm_rbm_timer_start(bm_rbm *h);
int bm_rbm_timer_stop(bm_rbm *h, float *out_ms);
int bm_rbm_timer_mark(bm_rbm *h);                       /* record the end event only (no host wait) ... */
int bm_rbm_timer_elapsed(bm_rbm *h, float *out_ms);     /* ... wait for it and read start -> mark */

/* ---- float64 RBM path ---------------------------------------------------------------------
 * The reference's dtype is a constructor argument (base/mixin.py:15, DtypeMixin) and its own
 * tests train a float64 BernoulliRBM (rbm/tests/test_rbm.py:53-56,70-73).  Same fetch sites as
 * the float32 entry points above, every buffer and scalar in double; Bernoulli or Gaussian visible units,
 * Bernoulli or Multinomial hidden units.  Variables as in
 * bm_rbm_set_param.  Compatibility path on the FP64 matrix cores, ~3x the float32 update. */
typedef struct bm_rbm64 bm_rbm64;
/* hyper5 = {l2, sparsity_target, sparsity_cost, sparsity_damping, dropout (<0: off)} as doubles (a Python
 * float is a double; the float fields of cfg would round them); NULL: take them from cfg */
int bm_rbm64_create(const bm_rbm_config *cfg, const double *hyper5, bm_rbm64 **out);
int bm_rbm64_destroy(bm_rbm64 *h);
int bm_rbm64_sync(bm_rbm64 *h);
int bm_rbm64_seed(bm_rbm64 *h, uint64_t seed);                                  /* tf_model.py:20-21 */
int bm_rbm64_set_row_offset(bm_rbm64 *h, int64_t row0);
int bm_rbm64_set_param(bm_rbm64 *h, const char *name, const double *host, size_t n);   /* tf_model.py:22-28 */
int bm_rbm64_get_param(bm_rbm64 *h, const char *name, double *host, size_t n);         /* tf_model.py:183-202 */
int bm_rbm64_train_step(bm_rbm64 *h, const double *X_dev, int32_t B,                    /* base_rbm.py:566 */
                        double learning_rate, double momentum, int32_t n_gibbs_steps);
int bm_rbm64_train_step_metrics(bm_rbm64 *h, const double *X_dev, int32_t B,            /* base_rbm.py:554-564 */
                                double learning_rate, double momentum, int32_t n_gibbs_steps, double *out4);
int bm_rbm64_transform(bm_rbm64 *h, const double *X_dev, int32_t B, int32_t n_gibbs_steps,   /* base_rbm.py:697 */
                       double *H_dev);
int bm_rbm64_metrics(bm_rbm64 *h, const double *X_dev, int32_t B, int32_t n_gibbs_steps,     /* base_rbm.py:578 */
                     double *out4);
int bm_rbm64_free_energy(bm_rbm64 *h, const double *X_dev, int32_t B, double *out1);          /* base_rbm.py:605,612 */


/* ------------------------------------------------------------------- DBM */
typedef struct bm_dbm bm_dbm;

#define BM_DBM_MAX_LAYERS 4

/* ctor kwargs of DBM.__init__ (dbm.py:89-99) + layer sizes (dbm.py:207-231). */
typedef struct bm_dbm_config {
    int32_t n_layers;                       /* number of hidden layers */
    int32_t n_visible;
    int32_t n_hiddens[BM_DBM_MAX_LAYERS];
    int32_t v_unit;                         /* BM_UNIT_* of the visible layer */
    int32_t sample_v_states;
    int32_t sample_h_states[BM_DBM_MAX_LAYERS];
    int32_t n_particles;                    /* M (dbm.py:254-255) */
    int32_t batch_size;                     /* N; mu variables are [batch_size, n_i] (dbm.py:345-348) */
    int32_t max_mf_updates;
    float   mf_tol;
    float   l2;
    float   max_norm;                       /* +inf => off (dbm.py:511-513) */
    float   sparsity_target[BM_DBM_MAX_LAYERS];
    float   sparsity_cost[BM_DBM_MAX_LAYERS];
    float   sparsity_damping;
    /* hidden layer kinds (layers.py:39-70; the CIFAR DBM of examples/dbm_cifar.py is Gaussian-Bernoulli-
     * Multinomial): BM_UNIT_BERNOULLI (0, default) or BM_UNIT_MULTINOMIAL with n_samples[i] draws per row */
    int32_t h_unit[BM_DBM_MAX_LAYERS];
    int32_t n_samples[BM_DBM_MAX_LAYERS];
} bm_dbm_config;

int bm_dbm_create(const bm_dbm_config *cfg, bm_dbm **out);
int bm_dbm_destroy(bm_dbm *h);
int bm_dbm_sync(bm_dbm *h);
int bm_dbm_seed(bm_dbm *h, uint64_t seed);
int bm_dbm_set_row_offset(bm_dbm *h, int64_t row0, int64_t particle0);

/* names: "W","W_1",.. "hb","hb_1",.. "vb", "dW*","dhb*","dvb", "mu*","q_means*",
 * "mu_means*", "v" (particles [M,V]), "h","h_1",.. ([M,n_i]), "sigma"
 * (get_tf_params naming, examples/dbm_mnist.py:367-371). */
int bm_dbm_set_param(bm_dbm *h, const char *name, const float *host, size_t n);
int bm_dbm_get_param(bm_dbm *h, const char *name, float *host, size_t n);
int bm_dbm_dev_ptr(bm_dbm *h, const char *name, void **out_dev, size_t *out_n);

/* session.run(train_op) at dbm.py:805 (graph dbm.py:515-621): mean-field on
 * X_dev [batch_size, V], n_gibbs_steps PCD sweeps on the particles, gradient
 * + sparsity + momentum + max-norm update.  out_n_mf (host, may be NULL)
 * receives the executed mean-field sweeps (n_mf_updates, dbm.py:631);
 * out_msre (host, may be NULL) the reconstruction msre (dbm.py:625-630). */
int bm_dbm_train_step(bm_dbm *h, const float *X_dev, float learning_rate, float momentum,
                      int32_t n_gibbs_steps, int32_t *out_n_mf, float *out_msre);
/* session.run([msre, n_mf_updates]) at dbm.py:813 (_run_val_metrics): the two tensors are built under
 * control dependencies on the mean-field AND the particle updates (dbm.py:521-523), so this fetch runs
 * the mean-field on X_dev, advances the fantasy particles by n_gibbs_steps, and returns the
 * reconstruction msre (dbm.py:625-630); no parameter update. */
int bm_dbm_metrics(bm_dbm *h, const float *X_dev, int32_t n_gibbs_steps, int32_t *out_n_mf, float *out_msre);
/* data-parallel halves, as for the RBM: phase 1 (mean-field, PCD, raw sums) leaves
 * [sum_b below^T mu_i | sum_m below^T H_i per layer | column sums of X, v, mu_i, H_i] in the
 * "grad" buffer (bm_dbm_dev_ptr); the caller all-reduces it; phase 2 applies the update of
 * dbm.py:550-621 with the GLOBAL batch size and particle count. */
int bm_dbm_grad_step(bm_dbm *h, const float *X_dev, int32_t n_gibbs_steps, int32_t *out_n_mf);
int bm_dbm_apply_step(bm_dbm *h, int32_t N_global, int32_t M_global,
                      float learning_rate, float momentum);
/* The mean-field loop condition (dbm.py:449-452) is a max over ALL rows of the minibatch:
 * under data parallelism the library calls `fn(local_max, ctx)` once per sweep and uses the
 * returned value (the caller implements it as an all-reduce(max) over ranks).  NULL = local. */
int bm_dbm_set_mf_allreduce(bm_dbm *h, float (*fn)(float local_max, void *ctx), void *ctx);
int bm_dbm_stream(bm_dbm *h, void **out_stream);

/* _make_mf (dbm.py:429-478) on X_dev [batch_size, V]; leaves mu in the
 * handle; copies the top layer's mu to MU_top_dev if non-NULL
 * (transform, dbm.py:859-872). */
int bm_dbm_mean_field(bm_dbm *h, const float *X_dev, float *MU_top_dev, int32_t *out_n_mf);
/* reconstruction op (dbm.py:625-633, public dbm.py:874-885): runs MF then
 * sigma(mu0 W0^T + vb) into R_dev [batch_size, V]. */
int bm_dbm_reconstruct(bm_dbm *h, const float *X_dev, float *R_dev);
/* sample_v op (dbm.py:641-648, public :887-897): k PCD sweeps, one mean
 * sweep, v <- v_means; copies v to V_dev [M, V] if non-NULL. */
int bm_dbm_sample_v(bm_dbm *h, int32_t n_gibbs_steps, float *V_dev);
/* AIS (dbm.py:650-736; public log_Z :899-939) for the 2-layer Bernoulli
 * DBM: n_runs chains, n_betas temperatures, n_gibbs_steps transitions per
 * temperature.  values_host [n_runs] receives the per-chain log Z estimates
 * (host post-processing with log_mean_exp stays in Python, dbm.py:935-939).
 * chain0 = global index of this rank's first chain (chains shard over ranks).
 * The per-beta terms are fp32 as in the reference; their sum over the betas is kept in DOUBLE on the device (the
 * reference accumulates it in fp32, dbm.py:708-728, and loses nats at 1000 betas): a deliberate, documented deviation. */
int bm_dbm_ais(bm_dbm *h, int32_t n_betas, int32_t n_runs, int32_t n_gibbs_steps,
               uint64_t seed, int64_t chain0, float *values_host);
/* log_proba op (dbm.py:738-759): MF then -E_q[E] + H(mu) per row of
 * X_dev [batch_size, V] into out_host [batch_size] (log Z is subtracted by
 * the Python caller, dbm.py:955-956). */
int bm_dbm_log_proba(bm_dbm *h, const float *X_dev, float *out_host);

int bm_dbm_timer_start(bm_dbm *h);
int bm_dbm_timer_stop(bm_dbm *h, float *out_ms);
int bm_dbm_timer_mark(bm_dbm *h);
int bm_dbm_timer_elapsed(bm_dbm *h, float *out_ms);


/* ---- RCCL inside the library (bm_comm_init / bm_allreduce_grads; §8e: one exchange step
 * per update).  The host only transports the 128-byte id from rank 0 to the other ranks.  librccl is
 * dlopen'ed at first use (BM355_RCCL_LIB overrides the name); the single-GPU path does not need it.
 * boltzmann_machines_amd/parallel.py can drive either these or torch.distributed (bench.py default). */
typedef struct bm_comm bm_comm;
int bm_comm_unique_id(void *out_id128);                                   /* ncclGetUniqueId, rank 0 */
int bm_comm_init(int32_t rank, int32_t nranks, const void *id128, bm_comm **out);   /* on the current device */
int bm_comm_destroy(bm_comm *c);
/* in-place all-reduce(sum) / all-gather of device floats, enqueued on `stream` (a hipStream_t) */
int bm_comm_allreduce_sum(bm_comm *c, float *buf_dev, size_t count, void *stream);
int bm_comm_allgather(bm_comm *c, const float *send_dev, float *recv_dev, size_t count_per_rank, void *stream);
int bm_comm_allreduce_max(bm_comm *c, float *buf_dev, size_t count, void *stream);
int bm_comm_rank(bm_comm *c, int32_t *out_rank, int32_t *out_nranks);
/* Data-parallel DBM (dbm.py:449-452 over the GLOBAL minibatch): with a communicator installed the mean-field
 * residual of every sweep is all-reduced (max) on the device, on the handle's stream, and the device-side
 * loop control latches the same `done` on every rank - no host callback, no host synchronisation per sweep.
 * NULL removes it.  (bm_dbm_set_mf_allreduce remains for collectives the library does not own.) */
int bm_dbm_set_comm(bm_dbm *h, bm_comm *c);
/* Chain-sharded AIS (north_star): this rank runs its contiguous slice of the n_runs_total chains
 * (the chain index in the RNG stream is global), zero communication during the sweep, ONE all-gather of the
 * per-chain values at the end; values_host [n_runs_total] is filled on every rank.  Replaces the
 * session.run(log_Z) of dbm.py:930 in a multi-GPU job. */
int bm_dbm_ais_sharded(bm_dbm *h, bm_comm *c, int32_t n_betas, int32_t n_runs_total, int32_t n_gibbs_steps,
                       uint64_t seed, float *values_host);
/* the exchange step of data-parallel training: all-reduce of the handle's fused "grad" buffer on the
 * handle's stream, between bm_*_grad_step and bm_*_apply_step (no host synchronisation) */
int bm_rbm_allreduce_grads(bm_rbm *h, bm_comm *c);
/* Delayed-gradient data parallelism - a NON-parity mode (the reference is synchronous; here the update of step t is
 * the reduced gradient of step t-1), for when the all-reduce must leave the critical path.  The handle has two
 * gradient slots: bm_rbm_set_grad_slot picks the one bm_rbm_grad_step writes and bm_rbm_apply_step reads;
 * bm_rbm_allreduce_grads_async reduces the current slot on the handle's communication stream, ordered after
 * everything enqueued so far on the compute stream, and returns at once; bm_rbm_wait_grads makes the compute stream
 * wait for the reduction of a slot.  Per step t: set_grad_slot(t % 2), grad_step, allreduce_grads_async; then, for
 * t > 0: wait_grads((t-1) % 2), set_grad_slot((t-1) % 2), apply_step. */
int bm_rbm_set_grad_slot(bm_rbm *h, int32_t slot);
int bm_rbm_allreduce_grads_async(bm_rbm *h, bm_comm *c);
int bm_rbm_wait_grads(bm_rbm *h, int32_t slot);
int bm_dbm_allreduce_grads(bm_dbm *h, bm_comm *c);

/* ---- one-shot exchange over peer-mapped device memory (SURVEY §5 "Distributed communication backend": a direct
 * reduce-scatter + all-gather over the 7 xGMI links instead of a ring; §8e: the ONE exchange step of a data-parallel
 * update).  One process per GPU.  Every rank: bm_xchg_create on the buffer it wants summed -> bm_xchg_export (a
 * 256-byte blob: hipIpc handles of the buffer, a staging slice and the flag words) -> the host gathers the blobs
 * of all ranks in rank order (any channel) -> bm_xchg_attach.  bm_xchg_allreduce_sum then runs ONE kernel per rank
 * on `stream`: rank r adds slice r of every rank's buffer in rank order 0..N-1 (16-byte cache-bypassing loads over
 * all links at once), publishes it, and pulls the other slices from their owners; every rank ends with the same
 * bits.  Every wait inside the kernel is bounded (BM_XCHG_TIMEOUT_S, default 20 s); bm_xchg_status reports a
 * timed-out wait instead of a hung device.  The RCCL path (bm_comm_*) computes the same sums in ring order. */
typedef struct bm_xchg bm_xchg;
int bm_xchg_create(int32_t rank, int32_t nranks, float *buf_dev, size_t count, bm_xchg **out);
int bm_xchg_blob_bytes(void);                                            /* 256 */
int bm_xchg_export(bm_xchg *x, void *out_blob256);
int bm_xchg_attach(bm_xchg *x, const void *all_blobs /* nranks x 256 bytes, ordered by rank */);
int bm_xchg_destroy(bm_xchg *x);
int bm_xchg_allreduce_sum(bm_xchg *x, void *stream);
/* all-reduce(max) of ONE non-negative float in this rank's device memory: one 8-byte {value, epoch} store into
 * every peer's slot, one fabric hop (the mean-field residual of a data-parallel DBM, dbm.py:449-452) */
int bm_xchg_allreduce_max1(bm_xchg *x, float *val_dev, void *stream);
int bm_xchg_status(bm_xchg *x, int32_t *out_status);                     /* synchronises; 0 = no wait timed out */
int bm_xchg_set_timeout(bm_xchg *x, double seconds);                     /* bound of the in-kernel waits of later launches */
/* at most n workgroups per exchange launch (default: up to one per CU); before the first exchange.  Only for ranks that
 * SHARE a device (dry runs): workgroups spinning on every CU can keep the peer process's kernels from being placed. */
int bm_xchg_set_max_workgroups(bm_xchg *x, int32_t n);
int bm_xchg_info(bm_xchg *x, int32_t *out_rank, int32_t *out_nranks, size_t *out_count);
/* the handle's fused "grad" buffer as the exchanged buffer; bm_*_allreduce_grads_direct is the drop-in for
 * bm_*_allreduce_grads between bm_*_grad_step and bm_*_apply_step */
int bm_rbm_xchg_create(bm_rbm *h, int32_t rank, int32_t nranks, bm_xchg **out);
int bm_dbm_xchg_create(bm_dbm *h, int32_t rank, int32_t nranks, bm_xchg **out);
int bm_rbm_allreduce_grads_direct(bm_rbm *h, bm_xchg *x);
int bm_dbm_allreduce_grads_direct(bm_dbm *h, bm_xchg *x);
/* Data-parallel CD-k, the exchange AND the parameter update in ONE launch (replaces bm_rbm_allreduce_grads_direct +
 * bm_rbm_apply_step; same bits): rank r sums slice r of the W part of every rank's gradient buffer in rank order,
 * applies g = raw / N - l2 W - pen, dW = lr (mom dW + g), W += dW (base_rbm.py:446-468) to ITS slice of W / dW and
 * every rank gathers the updated slices of W; the [V | H | H] tail is reduced by every rank itself (rank order) and
 * the bias / q_means update applied to every replica.  After it all replicas hold the same W, vb, hb, dvb, dhb,
 * q_means; of the momentum buffer dW a rank holds its own slice - bm_rbm_exchange_gather_dw completes the replicas
 * (call it before dW is read: checkpoints, get_param("dW")).  Needs n_hidden % 4 == 0. */
int bm_rbm_exchange_apply_direct(bm_rbm *h, bm_xchg *x, int32_t B_global, float learning_rate, float momentum);
int bm_rbm_exchange_gather_dw(bm_rbm *h, bm_xchg *x);
/* The same for the data-parallel DBM update (replaces bm_dbm_allreduce_grads_direct + bm_dbm_apply_step; same bits), with
 * COLUMN-sliced ownership because the update ends with the max-norm rescale of whole columns (dbm.py:511-513, 603-606):
 * rank r owns columns [r sw_i, (r + 1) sw_i) of every W_i (sw_i = 32 ceil(n_{i+1} / (32 nranks))).  One launch sums the
 * owned columns of the raw outer products over the ranks (rank order), applies the update to them and - every rank for
 * itself - the bias / q_means / mu_means / penalty update from the reduced column sums (dbm.py:550-600); the max-norm
 * kernels run on the owned columns; a second launch gathers the other ranks' columns into W_i, W_i^T and the column norms.
 * The momentum buffers dW_i stay with their owners: bm_dbm_get_param("dW"), bm_dbm_train_step and bm_dbm_apply_step fail
 * until bm_dbm_exchange_gather_dw.  Needs every hidden width to be a multiple of 4 and the exchange created by
 * bm_dbm_xchg_create for this engine (its staging slice is sized for the columns): bm_dbm_exchange_apply_ok says. */
int bm_dbm_exchange_apply_ok(bm_dbm *h, bm_xchg *x, int32_t *out_ok);
int bm_dbm_exchange_apply_direct(bm_dbm *h, bm_xchg *x, int32_t N_global, int32_t M_global, float learning_rate, float momentum);
int bm_dbm_exchange_gather_dw(bm_dbm *h, bm_xchg *x);
/* bm_dbm_ais_sharded over the direct exchange instead of the RCCL communicator (same slices, same values bit for bit): the
 * per-chain values travel through the exchange's registered buffer as an all-reduce(sum) of a window in which every rank
 * wrote its own chains and zeros elsewhere (x + 0 = x: every rank receives the owner's bits).  Overwrites the head of the
 * engine's gradient payload (recomputed by every update).  A rank whose sweep fails contributes NaN and every rank
 * reports the error; a rank that never arrives ends the wait at the exchange's time-out (sticky status). */
int bm_dbm_ais_sharded_direct(bm_dbm *h, bm_xchg *x, int32_t n_betas, int32_t n_runs_total, int32_t n_gibbs_steps,
                              uint64_t seed, float *values_host);
/* Opt-in "fast-binary" mode (SURVEY §7 hard part 4; csrc/bm_bf3.h): contractions whose input states are {0,1}
 * bitmaps (AIS with all layers sampled; the RBM sampling sweep with both layers sampled; in the PCD particle sweeps of
 * bm_dbm_train_step / bm_dbm_sample_v every contraction over a Bernoulli layer sampled earlier in the same call - a
 * real-valued visible layer and the particles a call starts from are read in fp32) split the fp32 weights
 * EXACTLY into three bf16 planes and run on the bf16 matrix cores - exact products, fp32 accumulation in a
 * different order than the default chain: results agree to fp32 round-off (free energy / log-weights 1e-5, bitmaps
 * identical except where |u - p| is at round-off distance), NOT bit for bit.  Never the default.
 * on = 1: where the mode was measured FASTER than the fp32 path - AIS always; the sampling sweep of an RBM and the particle
 * sweeps of a DBM only from 8M weights in the (bottom) weight matrix upwards (3072 x 5000 gains, 784 x 1024 loses: there the
 * passes are bound by their fill and epilogue, not by matrix time).  on = 2: wherever legal (tests, measurements). */
int bm_dbm_set_fast_binary(bm_dbm *h, int32_t on);
int bm_rbm_set_fast_binary(bm_rbm *h, int32_t on);
/* bm_dbm_ais accumulation.  0 (default): per chain the DIFFERENCE log p*_b(x) - log p*_a(x) of consecutive betas, its
 * row sums and the running log-weight in double, in a fixed order.  1: LITERALLY the reference's float32 arithmetic
 * (dbm.py:650-660, :708-728): each log p*_beta(x) formed in float32 and added to / subtracted from a float32
 * log-weight in the graph's order; two extra score-only passes per beta. */
int bm_dbm_set_ais_literal(bm_dbm *h, int32_t on);
/* Sigmoid of the Bernoulli layers.  0 (default): the engine's form (one correctly rounded division, e / (1 + e) for x < 0;
 * ~1.4 ulp).  1: LITERALLY tf.nn.sigmoid as the reference evaluates it (layers.py:47-48 -> float32 1 / (1 + exp(-x)), exp
 * as Eigen's pexp of TensorFlow 1.3; ~1.8 ulp) in every pass of this engine: the same values to float32 round-off, and the
 * reference's mean-field trip counts - the loop of dbm.py:449-452 at mf_tol = 1e-7 is decided in the last bits of the
 * means (784-512-1024: 5 - 6 sweeps per update with the literal form, 7 - 8 with the default). */
int bm_dbm_set_sigmoid_literal(bm_dbm *h, int32_t on);
/* ---- float64 DBM path ----------------------------------------------------------------------
 * DBM(dtype='float64') (base/mixin.py:14-25; the DBM graph is built "all in model dtype", dbm.py:294-383): the fetch sites
 * of the float32 entry points above with every buffer, hyper-parameter and random draw in double, on the FP64 tile engine of
 * the float64 RBM path (sequential ascending-k fma chains: bit-identical to oracle/bm_oracle_dbm64.c for everything that
 * feeds back into state).  A compatibility path (csrc/bm_dbm64.hip): Bernoulli hidden layers, Bernoulli or Gaussian visible
 * units, one process; the mean-field loop is driven by the host.  Variables as in bm_dbm_set_param.
 * hyper12 = {mf_tol, l2, max_norm, sparsity_damping, sparsity_target[4], sparsity_cost[4]} as doubles (NULL: from cfg). */
typedef struct bm_dbm64 bm_dbm64;
int bm_dbm64_create(const bm_dbm_config *cfg, const double *hyper12, bm_dbm64 **out);
int bm_dbm64_destroy(bm_dbm64 *h);
int bm_dbm64_sync(bm_dbm64 *h);
int bm_dbm64_seed(bm_dbm64 *h, uint64_t seed);                                           /* tf_model.py:20-21 */
int bm_dbm64_set_row_offset(bm_dbm64 *h, int64_t row0, int64_t particle0);
int bm_dbm64_set_param(bm_dbm64 *h, const char *name, const double *host, size_t n);     /* tf_model.py:22-28 */
int bm_dbm64_get_param(bm_dbm64 *h, const char *name, double *host, size_t n);           /* tf_model.py:183-202 */
int bm_dbm64_train_step(bm_dbm64 *h, const double *X_dev, double learning_rate, double momentum,   /* dbm.py:805 */
                        int32_t n_gibbs_steps, int32_t *out_n_mf, double *out_msre);
int bm_dbm64_metrics(bm_dbm64 *h, const double *X_dev, int32_t n_gibbs_steps, int32_t *out_n_mf, double *out_msre);   /* dbm.py:813 */
int bm_dbm64_mean_field(bm_dbm64 *h, const double *X_dev, double *out_dev, int32_t *out_n_mf);   /* dbm.py:866 */
int bm_dbm64_reconstruct(bm_dbm64 *h, const double *X_dev, double *R_dev);                        /* dbm.py:881 */
int bm_dbm64_sample_v(bm_dbm64 *h, int32_t n_gibbs_steps, double *V_dev);                         /* dbm.py:892 */
int bm_dbm64_ais(bm_dbm64 *h, int32_t n_betas, int32_t n_runs, int32_t n_gibbs_steps, uint64_t seed, int64_t chain0,
                 double *values_host);                                                            /* dbm.py:930 */
int bm_dbm64_log_proba(bm_dbm64 *h, const double *X_dev, double *out_host);                       /* dbm.py:953 */

/* like bm_dbm_set_comm, with the per-sweep residual max going through bm_xchg_allreduce_max1; NULL removes it */
int bm_dbm_set_xchg(bm_dbm *h, bm_xchg *x);

#ifdef __cplusplus
}
#endif
#endif /* BM355_H */
