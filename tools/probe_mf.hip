// tools/probe_mf.hip — ablation probe for the two MEAN-FIELD passes of the 784-512-1024 DBM at 512 rows (developer tool).
//   h1 pass: mu1 = sigmoid(X.W0 [stored partial, acc_init] + mu2.W1^T + hb0)   I = 512, J = 512, K = 1024, x-major P
//   h2 pass: mu2 = sigmoid(mu1.W1 + hb1)                                        I = 1024, J = 512, K = 512, x-major P
// exactly as bm_dbm.hip gibbs_sweep issues them (previous mu for the residual, per-workgroup residual slots, the loop
// control of the previous sweep evaluated by the h1 pass), timed as 200 back-to-back launches, with the compile-time
// ablation masks of bm_gemm.h and the cycle stamps of BM_PROBE.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DBM_PROBE tools/probe_mf.hip -o tools/probe_mf
#include "../boltzmann_machines_amd/csrc/bm_common.h"
#include "../boltzmann_machines_amd/csrc/bm_kernels.h"
#include <vector>
namespace bm { void set_error(const char *, ...) {} }
using namespace bm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_empty(float *o) { if (threadIdx.x == 999) o[0] = 1.f; }
template <class F> static float time_it(hipStream_t st, hipEvent_t e0, hipEvent_t e1, F f) {
    for (int i = 0; i < 20; ++i) f();
    (void)hipStreamSynchronize(st);
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < 200; ++i) f();
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return 1e3f * ms / 200;
}
int main() {
    const int H1 = 512, H2 = 1024, B = 512;
    Mat W1, W1t, Mu1, Mu2, Mu1b, Mu2b, XW0;
    W1.alloc(H1, H2); W1t.alloc(H2, H1); Mu1.alloc(B, H1); Mu2.alloc(B, H2); Mu1b.alloc(B, H1); Mu2b.alloc(B, H2); XW0.alloc(B, H1);
    std::vector<float> hw((size_t)H1 * H2), hm((size_t)B * H2);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.01f * (float)((i * 2654435761u >> 8) % 2001 - 1000) / 1000.f;
    for (size_t i = 0; i < hm.size(); ++i) hm[i] = (float)((i * 2246822519u >> 7) % 1000) / 1000.f;
    W1.upload(hw.data()); W1t.upload(hw.data()); Mu2.upload(hm.data()); Mu2b.upload(hm.data()); Mu1.upload(hm.data()); Mu1b.upload(hm.data()); XW0.upload(hm.data());
    float *hb; CK(hipMalloc((void **)&hb, 1024 * 4)); CK(hipMemset(hb, 0, 1024 * 4));
    float *slots; CK(hipMalloc((void **)&slots, 4 * BM_MF_SLOTS * 4 * 2)); CK(hipMemset(slots, 0, 4 * BM_MF_SLOTS * 4 * 2));
    MfCtl *ctl; CK(hipMalloc((void **)&ctl, sizeof(MfCtl))); CK(hipMemset(ctl, 0, sizeof(MfCtl)));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    long long *dbg; CK(hipMalloc((void **)&dbg, 8192 * 8)); CK(hipMemset(dbg, 0, 8192 * 8));
    TileMap tm = make_tile_map(1, 1, 1.0, 1.0, -1);
    // ---- the h1 pass
    ActArgs a1; memset(&a1, 0, sizeof(a1));
    a1.P1 = make_operand(W1.p, W1.ld, H1); a1.p_xm = 1;           // W_1 [i = h1][k = h2]
    a1.Q1 = make_operand(Mu2.p, Mu2.ld, B); a1.K1 = H2;
    a1.I = H1; a1.J = B; a1.bias = hb; a1.mult = 1.f; a1.bmult = 1.f; a1.kind = 0; a1.sample = 0;
    a1.means = Mu1b.p; a1.ldo = Mu1b.ld; a1.key = PhiloxKey{1, 2, 3, 4};
    a1.acc_init = XW0.p; a1.ld_init = XW0.ld;
    a1.prev = Mu1.p; a1.maxdiff = &ctl->maxdiff; a1.maxdiff_blk = slots;
    a1.skip = &ctl->done;
    a1.chk_ctl = ctl; a1.chk_slots = slots + 4 * BM_MF_SLOTS; a1.chk_n = 2 * BM_MF_SLOTS; a1.chk_tol = -1.f;    // never done
    a1.dbg = dbg;
    // ---- the h2 pass
    ActArgs a2; memset(&a2, 0, sizeof(a2));
    a2.P1 = make_operand(W1t.p, W1t.ld, H2); a2.p_xm = 1;         // W_1^T [i = h2][k = h1]
    a2.Q1 = make_operand(Mu1b.p, Mu1b.ld, B); a2.K1 = H1;
    a2.I = H2; a2.J = B; a2.bias = hb; a2.mult = 1.f; a2.bmult = 1.f; a2.kind = 0; a2.sample = 0;
    a2.means = Mu2b.p; a2.ldo = Mu2b.ld; a2.key = PhiloxKey{1, 2, 3, 4};
    a2.prev = Mu2.p; a2.maxdiff = &ctl->maxdiff; a2.maxdiff_blk = slots + BM_MF_SLOTS;
    a2.skip = &ctl->done;
    a2.dbg = dbg;
    printf("k_empty 256x256: %.2f us per launch (the kernel boundary)\n", time_it(st, e0, e1, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, hb); }));
#define RUNA(ARGS, GEO, MINB, MASK, DYN, NAME) { \
        const int nblk = tile_grid<GEO>(ARGS.I, ARGS.J); \
        const dim3 grid(nblk), blk(GEO::NT); \
        float us = time_it(st, e0, e1, [&] { hipLaunchKernelGGL((act_kernel<GEO, MINB, false, true, MASK, XM, STG_DMA, false, false, true>), grid, blk, DYN, st, ARGS, tm); }); \
        std::vector<long long> hd(8192); CK(hipMemcpy(hd.data(), dbg, 8192 * 8, hipMemcpyDeviceToHost)); \
        double d[4] = {0, 0, 0, 0}, s1 = 0, s2 = 0; const int nb = nblk < 256 ? nblk : 256; \
        for (int b = 0; b < nb; ++b) { for (int q = 0; q < 4; ++q) d[q] += hd[2048 + b * 8 + q + 1] - hd[2048 + b * 8 + q]; s1 += hd[b*4+1]-hd[b*4]; s2 += hd[b*4+2]-hd[b*4+1]; } \
        printf("%-52s %6.2f us | fill %5.0f sync %5.0f loop %6.0f tail %5.0f | start..loop end %6.0f epilogue %5.0f cycles (%d wgs x %d thr, lds %d KB + %d)\n", NAME, us, d[0]/nb, d[1]/nb, d[2]/nb, d[3]/nb, s1/nb, s2/nb, nblk, GEO::NT, GEO::SMEM_FLOATS * 4 / 1024, (int)(DYN) / 1024); }
    typedef Geo<2, 2, 1, 1, 64> GS;       // 32 x 32, 4 waves (the tuner's choice for the h1 pass, two workgroups per CU)
    typedef Geo<2, 2, 1, 1, 32> GS32;
    typedef Geo<2, 4, 1, 1, 64> G8;       // 32 x 64, 8 waves (the tuner's choice for the h2 pass)
    typedef Geo<4, 2, 1, 1, 64> G8b;      // 64 x 32, 8 waves
    printf("---- h1 pass: I = 512, J = 512, K = 1024 (+ stored X.W0), 0.537 GFLOP = 3.41 us at the fp32-MFMA peak\n");
    RUNA(a1, GS, 2, 0, 0, "(warm-up row: ignore)")
    RUNA(a1, GS, 2, 0, 0, "h1 as issued (32x32, one wg per CU)")
    { ActArgs b = a1; b.chk_ctl = nullptr; RUNA(b, GS, 2, 0, 0, "h1 without the loop-control check of the previous sweep") }
    { ActArgs b = a1; b.chk_ctl = nullptr; b.prev = nullptr; b.maxdiff = nullptr; b.maxdiff_blk = nullptr; RUNA(b, GS, 2, 0, 0, "h1 ... and without the residual") }
    { ActArgs b = a1; b.chk_ctl = nullptr; b.prev = nullptr; b.maxdiff = nullptr; b.maxdiff_blk = nullptr; b.acc_init = nullptr; b.skip = nullptr; RUNA(b, GS, 2, 0, 0, "h1 ... and without the stored partial / skip word") }
    RUNA(a1, GS, 2, 16, 0, "h1 no epilogue")
    RUNA(a1, GS, 2, 1, 0, "h1 no global -> LDS traffic")
    RUNA(a1, GS, 2, 45, 0, "h1 only MFMA (+ epilogue)")
    RUNA(a1, GS, 2, 63, 0, "h1 nothing")
    RUNA(a1, GS, 2, 0, 20 * 1024, "h1 32x32 with 20 KB of dynamic LDS: ONE wg per CU")
    RUNA(a1, GS, 2, 0, 1024, "h1 32x32 with 1 KB of dynamic LDS")
    RUNA(a1, GS, 2, 0, 60 * 1024, "h1 32x32 with 60 KB of dynamic LDS")
    RUNA(a1, GS, 1, 0, 20 * 1024, "h1 32x32, one wg per CU, register budget of one")
    RUNA(a1, GS32, 4, 0, 0, "h1 32x32 bk32 (4 wg/CU)")
    RUNA(a1, GS32, 2, 0, 52 * 1024, "h1 32x32 bk32, one wg per CU")
    RUNA(a1, G8, 1, 0, 0, "h1 32x64 8 waves (128 wgs)")
    RUNA(a1, G8b, 1, 0, 0, "h1 64x32 8 waves (128 wgs)")
    printf("---- h2 pass: I = 1024, J = 512, K = 512, 0.537 GFLOP = 3.41 us at the fp32-MFMA peak\n");
    RUNA(a2, G8, 1, 0, 0, "(warm-up row: ignore)")
    RUNA(a2, G8, 1, 0, 0, "h2 as issued (32x64, 8 waves, 1 wg/CU)")
    RUNA(a2, G8, 1, 0, 20 * 1024, "h2 as issued + 20 KB of dynamic LDS")
    RUNA(a2, G8, 1, 0, 40 * 1024, "h2 as issued + 40 KB of dynamic LDS")
    { ActArgs b = a2; b.prev = nullptr; b.maxdiff = nullptr; b.maxdiff_blk = nullptr; RUNA(b, G8, 1, 0, 0, "h2 without the residual") }
    RUNA(a2, G8, 1, 16, 0, "h2 no epilogue")
    RUNA(a2, G8, 1, 1, 0, "h2 no global -> LDS traffic")
    RUNA(a2, G8, 1, 45, 0, "h2 only MFMA (+ epilogue)")
    RUNA(a2, G8, 1, 63, 0, "h2 nothing")
    RUNA(a2, GS, 2, 0, 0, "h2 32x32 (512 wgs, 2 per CU)")
    RUNA(a2, G8b, 1, 0, 0, "h2 64x32 8 waves")
    // the pair as the loop issues it
    {
        const dim3 g1(tile_grid<GS>(a1.I, a1.J)), g2(tile_grid<G8>(a2.I, a2.J));
        ActArgs b1 = a1, b2 = a2; b1.dbg = nullptr; b2.dbg = nullptr;
        float us = time_it(st, e0, e1, [&] {
            hipLaunchKernelGGL((act_kernel<GS, 2, false, true, 0, XM, STG_DMA, false, false, true>), g1, dim3(GS::NT), 0, st, b1, tm);
            hipLaunchKernelGGL((act_kernel<G8, 1, false, true, 0, XM, STG_DMA, false, false, true>), g2, dim3(G8::NT), 0, st, b2, tm); });
        printf("one sweep = h1 + h2 back to back: %.2f us\n", us);
        us = time_it(st, e0, e1, [&] {
            hipLaunchKernelGGL((act_kernel<GS, 2, false, true, 0, XM, STG_DMA, false, false, true>), g1, dim3(GS::NT), 20 * 1024, st, b1, tm);
            hipLaunchKernelGGL((act_kernel<G8, 1, false, true, 0, XM, STG_DMA, false, false, true>), g2, dim3(G8::NT), 0, st, b2, tm); });
        printf("one sweep, h1 with one wg per CU:  %.2f us\n", us);
    }
    return 0;
}
