// tools/probe_act.hip — ablation probe for act_kernel (developer tool, not part of the product).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DBM_PROBE tools/probe_act.hip -o probe_act
#include "../boltzmann_machines_amd/csrc/bm_common.h"
#include "../boltzmann_machines_amd/csrc/bm_kernels.h"
#include <vector>
#include <algorithm>
namespace bm { void set_error(const char *, ...) {} }
using namespace bm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_empty(float *o) { if (threadIdx.x == 999) o[0] = 1.f; }
__global__ __launch_bounds__(256, 2) void k_lds(float *o) {
    __shared__ float sm[GeoAct::SMEM_FLOATS];
    sm[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (sm[(threadIdx.x + 1) & 255] == 999.f) o[0] = 1.f;
}
__global__ __launch_bounds__(256, 2) void k_args(ActArgs a) { if (threadIdx.x == 999) a.states[0] = a.mult; }
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
int main(int argc, char **argv) {
    const int V = 784, H = 1024, B = 512;
    Mat W, X, Hm, Hs;
    W.alloc(V, H); X.alloc(B, V); Hm.alloc(B, H); Hs.alloc(B, H);
    std::vector<float> hw((size_t)V * H), hx((size_t)B * V);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.01f * (float)((i * 2654435761u >> 8) % 2001 - 1000) / 1000.f;
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = ((i * 2246822519u >> 7) % 8) == 0 ? 1.f : 0.f;
    W.upload(hw.data()); X.upload(hx.data());
    float *hb; CK(hipMalloc((void **)&hb, H * 4)); CK(hipMemset(hb, 0, H * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    TileMap tm_slab = make_tile_map(1, 1, 1.0, 1.0, -1);      // the slab order (block_to_tile): no table
    ActArgs a; memset(&a, 0, sizeof(a));
    a.P1 = make_operand(W.p, W.ld, H); a.Q1 = make_operand(X.p, X.ld, B); a.K1 = V;
    a.I = H; a.J = B; a.bias = hb; a.mult = 1.f; a.kind = 0; a.sample = 1;
    a.means = Hm.p; a.states = Hs.p; a.ldo = Hm.ld; a.key = PhiloxKey{1, 2, 3, 4};
    printf("k_empty 256x256      %.2f us\n", time_it(st, e0, e1, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, Hs.p); }));
    printf("k_lds 73KB 256x256   %.2f us\n", time_it(st, e0, e1, [&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 0, st, Hs.p); }));
    printf("k_args 256x256       %.2f us\n", time_it(st, e0, e1, [&] { hipLaunchKernelGGL(k_args, dim3(256), dim3(256), 0, st, a); }));
    {   // does a HIP graph shrink the kernel boundary?  200 kernels per replay, stream launches vs graph replay
        auto graph_time = [&](auto enqueue, const char *name) {
            hipGraph_t gr; hipGraphExec_t ge;
            if (hipStreamBeginCapture(st, hipStreamCaptureModeGlobal) != hipSuccess) { printf("capture failed\n"); return; }
            for (int i = 0; i < 200; ++i) enqueue();
            if (hipStreamEndCapture(st, &gr) != hipSuccess) { printf("end capture failed\n"); return; }
            if (hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return; }
            for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge, st);
            (void)hipStreamSynchronize(st);
            (void)hipEventRecord(e0, st);
            for (int i = 0; i < 5; ++i) (void)hipGraphLaunch(ge, st);
            (void)hipEventRecord(e1, st);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%-34s %6.2f us per kernel in a 200-kernel graph\n", name, 1e3f * ms / 1000);
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(gr);
        };
        graph_time([&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, Hs.p); }, "k_empty (graph)");
        graph_time([&] { hipLaunchKernelGGL((act_kernel<GeoAct8, 1, false, true, 0>), dim3(tile_grid<GeoAct8>(a.I, a.J)), dim3(512), 0, st, a, tm_slab); }, "8w full (graph)");
        printf("%-34s %6.2f us per kernel, stream launches\n", "8w full (stream)",
               time_it(st, e0, e1, [&] { hipLaunchKernelGGL((act_kernel<GeoAct8, 1, false, true, 0>), dim3(tile_grid<GeoAct8>(a.I, a.J)), dim3(512), 0, st, a, tm_slab); }));
    }
    long long *dbg; CK(hipMalloc((void **)&dbg, 8192 * 8)); CK(hipMemset(dbg, 0, 8192 * 8));
    a.dbg = dbg;
#define RUNG(GEO, MINB, MASK, NAME) { \
        const int nblk = tile_grid<GEO>(a.I, a.J); \
        const dim3 grid(nblk), blk(GEO::NT); \
        float us = time_it(st, e0, e1, [&] { hipLaunchKernelGGL((act_kernel<GEO, MINB, false, true, MASK>), grid, blk, 0, st, a, tm_slab); }); \
        std::vector<long long> hd(8192); CK(hipMemcpy(hd.data(), dbg, 8192 * 8, hipMemcpyDeviceToHost)); \
        double d[4] = {0, 0, 0, 0}, s1 = 0, s2 = 0; \
        for (int b = 0; b < nblk; ++b) { for (int q = 0; q < 4; ++q) d[q] += hd[2048 + b * 8 + q + 1] - hd[2048 + b * 8 + q]; s1 += hd[b*4+1]-hd[b*4]; s2 += hd[b*4+2]-hd[b*4+1]; } \
        printf("%-34s %6.2f us | fill %5.0f  sync %5.0f  loop %6.0f  tail %5.0f | mainloop %6.0f epilogue %5.0f  (%d wgs x %d thr, lds %d KB)\n", NAME, us, d[0]/nblk, d[1]/nblk, d[2]/nblk, d[3]/nblk, s1/nblk, s2/nblk, nblk, GEO::NT, GEO::SMEM_FLOATS * 4 / 1024); }
#define RUN(MASK, NAME) RUNG(GeoAct, 1, MASK, NAME)
    RUN(0, "full")
    RUN(64, "full, careful steps only")
    RUN(16, "no-epilogue")
    RUN(1, "no-gload")
    RUN(2, "no-mfma")
    RUN(4, "no-ldswrite")
    RUN(8, "no-ldsread")
    RUN(32, "no-barrier")
    RUN(45, "only mfma (+epilogue)")
    RUN(63, "nothing")
    RUN(37, "mfma + reads")
    RUN(5, "mfma + reads + barrier")
    RUN(33, "mfma + reads + writes")
    RUN(9, "mfma + writes + barrier")
    RUN(4, "mfma + reads + gloads + barrier")
    printf("---- 8-wave geometry (the one the tuner picks at this shape)\n");
    RUNG(GeoAct8, 1, 0, "8w full")
    RUNG(GeoAct8, 1, 256, "8w no wait for the DMA")
    RUNG(GeoAct8, 1, 512, "8w HALF the Q bytes moved")
    {   // the same for the x-major-P instantiation (prop-down / prop-up from W^T): P = an [I][K] matrix
        Mat Wx; Wx.alloc(H, V);
        ActArgs ax = a; ax.P1 = make_operand(Wx.p, Wx.ld, H); ax.p_xm = 1;
        const dim3 grid(tile_grid<GeoAct8>(ax.I, ax.J)), blk(512);
        printf("%-34s %6.2f us\n", "8w x-major P full", time_it(st, e0, e1, [&] { hipLaunchKernelGGL((act_kernel<GeoAct8, 1, false, true, 0, XM>), grid, blk, 0, st, ax, tm_slab); }));
        printf("%-34s %6.2f us\n", "8w x-major P, HALF the Q bytes", time_it(st, e0, e1, [&] { hipLaunchKernelGGL((act_kernel<GeoAct8, 1, false, true, 512, XM>), grid, blk, 0, st, ax, tm_slab); }));
        printf("%-34s %6.2f us\n", "8w x-major P, no gload at all", time_it(st, e0, e1, [&] { hipLaunchKernelGGL((act_kernel<GeoAct8, 1, false, true, 1, XM>), grid, blk, 0, st, ax, tm_slab); }));
    }
    RUNG(GeoAct8, 1, 129, "8w lock-step no-gload")
    RUNG(GeoAct8, 1, 136, "8w lock-step no-ldsread")
    RUNG(GeoAct8, 1, 16, "8w no-epilogue")
    RUNG(GeoAct8, 1, 1, "8w no-gload")
    RUNG(GeoAct8, 1, 4, "8w no-ldswrite")
    RUNG(GeoAct8, 1, 5, "8w no-gload no-ldswrite")
    RUNG(GeoAct8, 1, 8, "8w no-ldsread")
    RUNG(GeoAct8, 1, 13, "8w no-gload/write/read")
    RUNG(GeoAct8, 1, 32, "8w no-barrier")
    RUNG(GeoAct8, 1, 45, "8w only mfma (+epilogue)")
    RUNG(GeoAct8, 1, 2, "8w no-mfma")
    RUNG(GeoAct8, 1, 63, "8w nothing")
    {   // phase-ordered step with stamps (wave 0 and wave 3 of every workgroup, step 4)
        RUN(0, "(phase-ordered probe removed)")
        std::vector<long long> hd(8192); CK(hipMemcpy(hd.data(), dbg, 8192 * 8, hipMemcpyDeviceToHost));
        for (int wv = 0; wv < 2; ++wv) {
            double d[5] = {0, 0, 0, 0, 0};
            for (int b = 0; b < 256; ++b) for (int q = 0; q < 5; ++q) d[q] += hd[4096 + b * 16 + wv * 8 + q + 1] - hd[4096 + b * 16 + wv * 8 + q];
            printf("   wave %d step 4: reads %5.0f  stores %5.0f  gloads %5.0f  mfmas %5.0f  barrier %5.0f cycles\n", wv * 3, d[0]/256, d[1]/256, d[2]/256, d[3]/256, d[4]/256);
        }
    }
    {   // ---- alternative geometries (same problem)
        typedef Geo<4, 2, 1, 1, 64> GA;      // 8 waves, 64 x 32 tile, wave 16 x 16
        typedef Geo<2, 4, 1, 1, 64> GA2;     // 8 waves, 32 x 64 tile
        typedef Geo<2, 2, 1, 1, 64> GS;      // 4 waves, 32 x 32 tile, 2 workgroups per CU
        typedef Geo<2, 2, 1, 1, 32> GS32;    // ... BK = 32
        typedef Geo<2, 2, 2, 1, 32> GB32;    // baseline tile, BK = 32
        RUNG(GA, 1, 0, "8w 64x32 bk64 full")
        RUNG(GA, 1, 45, "8w 64x32 bk64 only-mfma")
        RUNG(GA2, 1, 0, "8w 32x64 bk64 full")
        RUNG(GS, 2, 0, "4w 32x32 bk64 x2/CU full")
        RUNG(GS, 2, 45, "4w 32x32 bk64 x2/CU only-mfma")
        RUNG(GS32, 2, 0, "4w 32x32 bk32 x2/CU full")
        RUNG(GS32, 4, 0, "4w 32x32 bk32 (minb 4) full")
        RUNG(GB32, 1, 0, "4w 64x32 bk32 full")
        // two half-batch launches on two streams (independent Gibbs chains of rows 0-255 / 256-511)
        hipStream_t st2; CK(hipStreamCreate(&st2));
        hipEvent_t f0, f1; CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
        ActArgs h1 = a, h2 = a;
        h1.J = B / 2; h1.Q1 = make_operand(X.p, X.ld, B / 2); h1.dbg = nullptr;
        h2.J = B / 2; h2.Q1 = make_operand(X.p + (size_t)(B / 2) * X.ld, X.ld, B / 2); h2.row0 = B / 2;
        h2.means = Hm.p + (size_t)(B / 2) * Hm.ld; h2.states = Hs.p + (size_t)(B / 2) * Hs.ld; h2.dbg = nullptr;
#define RUN2(GEO, MINB, NAME, CHAIN) { \
            const dim3 grid(tile_grid<GEO>(h1.I, h1.J)), blk(GEO::NT); \
            auto once = [&] { \
                (void)hipEventRecord(f0, st); (void)hipStreamWaitEvent(st2, f0, 0); \
                for (int c = 0; c < CHAIN; ++c) { \
                    hipLaunchKernelGGL((act_kernel<GEO, MINB, false, true, 0>), grid, blk, 0, st, h1, tm_slab); \
                    hipLaunchKernelGGL((act_kernel<GEO, MINB, false, true, 0>), grid, blk, 0, st2, h2, tm_slab); } \
                (void)hipEventRecord(f1, st2); (void)hipStreamWaitEvent(st, f1, 0); }; \
            float us = time_it(st, e0, e1, once); \
            printf("%-34s %6.2f us per fork-join of %d kernels per stream = %6.2f us per full-batch kernel\n", NAME, us, CHAIN, us / CHAIN); }
        RUN2(GS, 2, "2 streams 4w 32x32 bk64", 1)
        RUN2(GS, 2, "2 streams 4w 32x32 bk64", 3)
        RUN2(GS32, 2, "2 streams 4w 32x32 bk32", 3)
        RUN2(GB32, 2, "2 streams 4w 64x32 bk32 (128 wgs each)", 3)
        RUN2(GeoAct, 1, "2 streams baseline geo (128 wgs each)", 3)
    }
    {   // ---- grad kernel (RBM form 0, fused update) with and without the bias groups
        Mat dW, Wt, Vs, Hk;
        dW.alloc(V, H); Wt.alloc(H, V); Vs.alloc(B, V); Hk.alloc(B, H);
        Vs.upload(hx.data());
        float *pen; CK(hipMalloc((void **)&pen, H * 4)); CK(hipMemset(pen, 0, H * 4));
        GradArgs g; memset(&g, 0, sizeof(g));
        g.Ppos = make_operand(Hm.p, Hm.ld, H); g.Qpos = make_operand(X.p, X.ld, V); g.Kpos = B;
        g.Pneg = make_operand(Hk.p, Hk.ld, H); g.Qneg = make_operand(Vs.p, Vs.ld, V); g.Kneg = B;
        g.I = H; g.J = V; g.form = 0; g.fused = 1; g.W = W.p; g.dW = dW.p; g.Wt = Wt.p; g.ldw = W.ld; g.ldwt = Wt.ld;
        g.pen = pen; g.N = B; g.M = B; g.l2 = 1e-5f; g.lr = 0.f; g.mom = 0.9f; g.dbg = dbg;
        {   // standalone bias/colsum groups
            RbmBiasFusedArgs bfa; memset(&bfa, 0, sizeof(bfa));
            float *vec6; CK(hipMalloc((void **)&vec6, 8 * (V + H) * 4)); CK(hipMemset(vec6, 0, 8 * (V + H) * 4));
            bfa.X = X.p; bfa.ldx = X.ld; bfa.vs = Vs.p; bfa.ldv = Vs.ld; bfa.h0m = Hm.p; bfa.ldh0 = Hm.ld; bfa.hm = Hk.p; bfa.ldh = Hk.ld; bfa.B = B;
            bfa.raw_tail = vec6;
            bfa.u.vb = vec6 + 2 * (V + H); bfa.u.dvb = bfa.u.vb + V; bfa.u.hb = bfa.u.dvb + V; bfa.u.dhb = bfa.u.hb + H; bfa.u.q = bfa.u.dhb + H; bfa.u.pen = bfa.u.q + H;
            bfa.u.V = V; bfa.u.H = H; bfa.u.N = B; bfa.u.lr = 0.f; bfa.u.mom = 0.f; bfa.u.damping = 0.9f;
            const int nw = (V + 63) / 64 + (H + 63) / 64;
            printf("rbm_bias_fused_kernel %d groups alone: %.2f us\n", nw, time_it(st, e0, e1, [&] { hipLaunchKernelGGL(rbm_bias_fused_kernel, dim3(nw), dim3(NT), 0, st, bfa); }));
            g.nbias = nw; g.bias = bfa;
            printf("grad + bias groups in one launch: %.2f us\n", time_it(st, e0, e1, [&] { launch_grad(g, st); }));
            g.nbias = 0;
        }
#define GRUN(MASK, NAME) { \
            const dim3 gg(tile_grid<GeoGrad>(g.I, g.J)); \
            float us = time_it(st, e0, e1, [&] { hipLaunchKernelGGL((grad_kernel<GeoGrad, true, MASK>), gg, dim3(NT), 0, st, g, tm_slab); }); \
            std::vector<long long> hd(8192); CK(hipMemcpy(hd.data(), dbg, 8192 * 8, hipMemcpyDeviceToHost)); \
            double s1 = 0, s2 = 0; for (int b = 0; b < 208; ++b) { s1 += hd[b*4+1]-hd[b*4]; s2 += hd[b*4+2]-hd[b*4+1]; } \
            printf("grad %-26s %6.2f us | mainloop %6.0f epilogue %5.0f\n", NAME, us, s1/208, s2/208); }
        GRUN(0, "full")
        GRUN(64, "full, careful steps only")
        GRUN(1, "no-gload")
        GRUN(2, "no-mfma")
        GRUN(4, "no-ldswrite")
        GRUN(8, "no-ldsread")
        GRUN(32, "no-barrier")
        GRUN(45, "only mfma")
        {   const dim3 gg4(tile_grid<GeoGrad>(g.I, g.J));
            printf("grad 4w REG staging              %6.2f us\n", time_it(st, e0, e1, [&] { hipLaunchKernelGGL((grad_kernel<GeoGrad, true, 0, STG_REG>), gg4, dim3(256), 0, st, g, tm_slab); }));
            printf("grad 8w REG staging              %6.2f us\n", time_it(st, e0, e1, [&] { hipLaunchKernelGGL((grad_kernel<GeoGrad8, true, 0, STG_REG>), gg4, dim3(512), 0, st, g, tm_slab); }));
            printf("act  8w REG staging              %6.2f us\n", time_it(st, e0, e1, [&] { hipLaunchKernelGGL((act_kernel<GeoAct8, 1, false, true, 0, KM, STG_REG>), dim3(tile_grid<GeoAct8>(a.I, a.J)), dim3(512), 0, st, a, tm_slab); }));
        }
        {   const dim3 gg(tile_grid<GeoGrad8>(g.I, g.J));
            printf("grad 8w ping-pong                %6.2f us\n", time_it(st, e0, e1, [&] { hipLaunchKernelGGL((grad_kernel<GeoGrad8, true, 0>), gg, dim3(512), 0, st, g, tm_slab); }));
            printf("grad 8w lock-step                %6.2f us\n", time_it(st, e0, e1, [&] { hipLaunchKernelGGL((grad_kernel<GeoGrad8, true, 128>), gg, dim3(512), 0, st, g, tm_slab); }));
            printf("grad 8w ping-pong no-gload       %6.2f us\n", time_it(st, e0, e1, [&] { hipLaunchKernelGGL((grad_kernel<GeoGrad8, true, 1>), gg, dim3(512), 0, st, g, tm_slab); }));
            printf("grad 8w ping-pong only-mfma      %6.2f us\n", time_it(st, e0, e1, [&] { hipLaunchKernelGGL((grad_kernel<GeoGrad8, true, 45>), gg, dim3(512), 0, st, g, tm_slab); }));
        }
        for (int variant = 0; variant < 1; ++variant) {
            g.Wt = (variant == 2) ? nullptr : Wt.p;
            float us = time_it(st, e0, e1, [&] { launch_grad(g, st); });
            std::vector<long long> hd(8192); CK(hipMemcpy(hd.data(), dbg, 8192 * 8, hipMemcpyDeviceToHost));
            double s1 = 0, s2 = 0; int nb = 208;
            for (int b = 0; b < nb; ++b) { s1 += hd[b*4+1]-hd[b*4]; s2 += hd[b*4+2]-hd[b*4+1]; }
            printf("grad %-22s %6.2f us | mainloop %6.0f epilogue %5.0f cycles\n", variant == 0 ? "tiles only" : variant == 1 ? "(repeat)" : "no Wt write", us, s1/nb, s2/nb);
        }
    }
    return 0;
}
