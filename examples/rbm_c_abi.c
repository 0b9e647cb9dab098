/* examples/rbm_c_abi.c - libbm355.so from plain C: the drop-in boundary has no Python, no torch and no C++ in it.
 * Trains a small BernoulliRBM for a few CD-1 updates on synthetic data and prints a checksum of the parameters
 * (tests/test_c_abi_program.py compares it with the same run through the ctypes binding).
 *   gcc -std=c99 -I include examples/rbm_c_abi.c -L boltzmann_machines_amd -lbm355 -Wl,-rpath,$PWD/boltzmann_machines_amd -o rbm_c_abi
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bm355.h"

#define CHECK(call) do { if ((call) != 0) { fprintf(stderr, "%s failed: %s\n", #call, bm_last_error()); return 1; } } while (0)

int main(void) {
    enum { V = 96, H = 72, B = 48, STEPS = 3 };
    if (bm_device_count() < 1) { fprintf(stderr, "no HIP device: libbm355 has no CPU fallback\n"); return 2; }
    bm_rbm_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.n_visible = V; cfg.n_hidden = H; cfg.max_batch = B;
    cfg.sample_v_states = 1; cfg.sample_h_states = 1;
    cfg.l2 = 1e-4f; cfg.dropout = -1.0f; cfg.sparsity_damping = 0.9f; cfg.sparsity_target = 0.1f;
    bm_rbm *h = NULL;
    CHECK(bm_rbm_create(&cfg, &h));
    /* deterministic inputs from a tiny LCG (the Python side repeats it) */
    static float W[V * H], X[B * V], Wout[V * H];
    uint32_t s = 12345u;
    for (int i = 0; i < V * H; ++i) { s = s * 1664525u + 1013904223u; W[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * 0.2f; }
    for (int i = 0; i < B * V; ++i) { s = s * 1664525u + 1013904223u; X[i] = (s >> 8) % 10u < 3u ? 1.0f : 0.0f; }
    CHECK(bm_rbm_set_param(h, "W", W, (size_t)V * H));
    CHECK(bm_rbm_seed(h, 2024u));
    void *Xd = NULL;
    CHECK(bm_dev_alloc(sizeof(X), &Xd));
    CHECK(bm_h2d(Xd, X, sizeof(X)));
    for (int t = 0; t < STEPS; ++t) CHECK(bm_rbm_train_step(h, (const float *)Xd, B, 0.05f, 0.5f, 1));
    CHECK(bm_rbm_get_param(h, "W", Wout, (size_t)V * H));
    /* order-dependent checksum of the bit patterns */
    uint64_t sum = 1469598103934665603ull;
    for (int i = 0; i < V * H; ++i) { uint32_t b; memcpy(&b, &Wout[i], 4); sum = (sum ^ b) * 1099511628211ull; }
    printf("W_FNV1A %016llx\n", (unsigned long long)sum);
    CHECK(bm_dev_free(Xd));
    CHECK(bm_rbm_destroy(h));
    return 0;
}
