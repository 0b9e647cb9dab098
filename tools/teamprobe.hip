// tools/teamprobe.hip — building blocks of a PERSISTENT update kernel (developer tool; profiles/NOTES.md §3.11):
//   (A) what one `buffer_inv sc1` / `buffer_wbl2 sc1` costs when every workgroup issues it (the agent-scope fences of
//       tools/chainprobe.hip cost 30 us per phase);
//   (B) XCD-local teams: the 32 workgroups that share an L2 (identified by HW_REG_XCC_ID + a ticket, not by blockIdx)
//       hand 8 KB each to all team mates through the L2, phase after phase over two ping-pong buffers; store / load cache
//       policies are template parameters, every value is checked, the time per phase is the hand-over + a 256 KB read;
//   (C) the same with readers on ALL XCDs (the outer products read every row block).
// Spins are bounded: no hang.     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/teamprobe.hip -o tools/teamprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NB = 256, TEAM = 32, WORDS = 2048;          // 8 KB per workgroup and phase

// cache policy codes: 0 plain, 1 sc0, 2 sc1, 3 sc0 sc1, 4 nt
// four loads + ONE wait in a single statement (the compiler does not track asm loads)
#define LD4X4(POLSTR) asm volatile("global_load_dwordx4 %0, %4, off " POLSTR "\n\tglobal_load_dwordx4 %1, %5, off " POLSTR "\n\t" \
                                   "global_load_dwordx4 %2, %6, off " POLSTR "\n\tglobal_load_dwordx4 %3, %7, off " POLSTR "\n\ts_waitcnt vmcnt(0)" \
                                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory")
template <int POL> __device__ __forceinline__ void ld4x4(f32x4 (&v)[4], const float *p0, const float *p1, const float *p2, const float *p3) {
    if (POL == 0) LD4X4("");
    if (POL == 1) LD4X4("sc0");
    if (POL == 2) LD4X4("sc1");
    if (POL == 3) LD4X4("sc0 sc1");
    if (POL == 4) LD4X4("nt");
}
template <int POL> __device__ __forceinline__ void st4(float *p, f32x4 v) {
    if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned ld_flag(const unsigned *p, int pol) {
    unsigned v;
    if (pol == 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_flag(unsigned *p, unsigned v, int pol) {
    if (pol == 2) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

// ---- (A)
__global__ __launch_bounds__(512) void k_fence(int what, int reps, int only_one, float *buf) {
    __shared__ float big[30 * 1024];
    big[threadIdx.x] = 0.f;
    // a little dirty data per round so that the writeback has something to do
    for (int r = 0; r < reps; ++r) {
        buf[(size_t)blockIdx.x * 2048 + threadIdx.x] = (float)r;
        if (threadIdx.x == 0 && (!only_one || blockIdx.x < 8)) {
            if (what & 1) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            if (what & 2) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
}

// ---- (B) / (C)
struct Team { unsigned ticket[8 * 32]; };    // 128-byte spacing
template <int ST, int LD, bool CROSS>
__global__ __launch_bounds__(512) void k_team(unsigned *ticket, unsigned *flags, float *buf0, float *buf1, int phases, unsigned gen0,
                                              int flag_pol, int *err, int *xcc_out, int work) {
    __shared__ float big[30 * 1024];
    __shared__ int s_slot, s_xcc, s_dead;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        s_xcc = (int)(id & 0xf); s_dead = 0;
        s_slot = (int)(__hip_atomic_fetch_add(ticket + s_xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) % TEAM);
        xcc_out[blockIdx.x] = s_xcc;
    }
    __syncthreads();
    const int xcc = s_xcc, slot = s_slot;
    if (xcc >= 8) { atomicExch(err, 2); return; }
    const int me = xcc * TEAM + slot;                    // logical workgroup: its data and its flag
    int bad = 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int p = 0; p < phases; ++p) {
        float *dst = (p & 1) ? buf1 : buf0;
        const unsigned gen = gen0 + p;
        // "work"
        const long long c0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - c0 < work) big[threadIdx.x] += 1.f;
        const float val = (float)(me * 7) + (float)gen;
        for (int i = threadIdx.x * 4; i < WORDS; i += blockDim.x * 4) st4<ST>(dst + (size_t)me * WORDS + i, (f32x4){val, val, val, val});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) st_flag(flags + me, gen, flag_pol);
        // wait for the team (CROSS: for everybody): wave 0 polls, one flag per lane
        if (w == 0) {
            unsigned spins = 0;
            const int nfl = CROSS ? NB : TEAM, base = CROSS ? 0 : xcc * TEAM;
            for (;;) {
                bool ok = true;
                for (int f = lane; f < nfl; f += 64) ok = ok && (ld_flag(flags + base + f, flag_pol) >= gen);
                if (__all(ok)) break;
                if (++spins > (1u << 14)) { if (lane == 0) { atomicExch(err, 1); s_dead = 1; } break; }
            }
        }
        __syncthreads();
        if (s_dead) return;                               // a wait expired: give up (bounded run time)
        // read the team's 256 KB (CROSS: 8 KB of one workgroup of every other team as well)
        const int i = threadIdx.x * 4;                   // WORDS == 4 * blockDim.x: one float4 per thread and source workgroup
        for (int q = 0; q < TEAM; q += 4) {
            f32x4 v[4];
            const int o = xcc * TEAM + q;
            ld4x4<LD>(v, dst + (size_t)o * WORDS + i, dst + (size_t)(o + 1) * WORDS + i, dst + (size_t)(o + 2) * WORDS + i, dst + (size_t)(o + 3) * WORDS + i);
            for (int u = 0; u < 4; ++u) {
                const float expect = (float)((o + u) * 7) + (float)gen;
                bad += (v[u][0] != expect) + (v[u][1] != expect) + (v[u][2] != expect) + (v[u][3] != expect);
            }
        }
        if (CROSS) for (int x = 0; x < 8; x += 4) {
            f32x4 v[4];
            const int s5 = (slot + 5) % TEAM;
            ld4x4<LD>(v, dst + (size_t)(x * TEAM + s5) * WORDS + i, dst + (size_t)((x + 1) * TEAM + s5) * WORDS + i,
                      dst + (size_t)((x + 2) * TEAM + s5) * WORDS + i, dst + (size_t)((x + 3) * TEAM + s5) * WORDS + i);
            for (int u = 0; u < 4; ++u) {
                const float expect = (float)(((x + u) * TEAM + s5) * 7) + (float)gen;
                bad += (v[u][0] != expect) + (v[u][1] != expect) + (v[u][2] != expect) + (v[u][3] != expect);
            }
        }
    }
    if (bad) atomicAdd(err + 1, bad);
}

template <int ST, int LD, bool CROSS>
static int run_team(const char *what, int flag_pol, int work) {
    const int PH = 200;
    unsigned *ticket, *flags; float *buf0, *buf1; int *err, *xcc;
    CK(hipMalloc((void **)&ticket, 8 * 32 * 4)); CK(hipMalloc((void **)&flags, NB * 4));
    CK(hipMalloc((void **)&buf0, (size_t)NB * WORDS * 4)); CK(hipMalloc((void **)&buf1, (size_t)NB * WORDS * 4));
    CK(hipMalloc((void **)&err, 8)); CK(hipMalloc((void **)&xcc, NB * 4));
    CK(hipMemset(ticket, 0, 8 * 32 * 4)); CK(hipMemset(flags, 0, NB * 4)); CK(hipMemset(err, 0, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_team<ST, LD, CROSS>), dim3(NB), dim3(512), 0, 0, ticket, flags, buf0, buf1, PH, 1u + rep * PH, flag_pol, err, xcc, work);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        { int h0; CK(hipMemcpy(&h0, err, 4, hipMemcpyDeviceToHost)); if (h0) break; }
    }
    int herr[2]; CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
    std::vector<int> hx(NB); CK(hipMemcpy(hx.data(), xcc, NB * 4, hipMemcpyDeviceToHost));
    int cnt[16] = {0}, mism = 0;
    for (int b = 0; b < NB; ++b) { cnt[hx[b] & 15]++; if (hx[b] != hx[b % 8]) ++mism; }
    printf("%-58s flags %s, work %5d: %6.2f us per phase; timeout/err %d, wrong values %d; per-XCC workgroups %d %d %d %d %d %d %d %d, not on blockIdx%%8's XCC: %d\n",
           what, flag_pol == 2 ? "sc1    " : "sc0 sc1", work, best * 1e3 / PH, herr[0], herr[1], cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5], cnt[6], cnt[7], mism);
    hipFree(ticket); hipFree(flags); hipFree(buf0); hipFree(buf1); hipFree(err); hipFree(xcc);
    return 0;
}

int main() {
    float *buf; CK(hipMalloc((void **)&buf, (size_t)NB * 2048 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int one = 0; one < 2; ++one)
        for (int what = 0; what < 4; ++what) {
            const int reps = 200;
            hipLaunchKernelGGL(k_fence, dim3(NB), dim3(512), 0, 0, what, 10, one, buf);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_fence, dim3(NB), dim3(512), 0, 0, what, reps, one, buf);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("(A) %s issued by %s: %.2f us per round (2 KB written per workgroup and round)\n",
                   what == 0 ? "nothing            " : what == 1 ? "buffer_wbl2 sc1    " : what == 2 ? "buffer_inv sc1     " : "wbl2 sc1 + inv sc1 ",
                   one ? "one workgroup per XCD " : "every workgroup (256) ", ms * 1e3 / reps);
        }
    // (B) team hand-over through the L2: store policy x load policy
    run_team<0, 2, false>("(B) team: plain stores, sc1 loads", 2, 0);
    run_team<0, 2, false>("(B) team: plain stores, sc1 loads", 3, 0);
    run_team<4, 2, false>("(B) team: nt stores, sc1 loads", 2, 0);
    run_team<3, 2, false>("(B) team: sc0 sc1 stores, sc1 loads", 2, 0);
    run_team<3, 2, false>("(B) team: sc0 sc1 stores, sc1 loads", 3, 0);
    run_team<2, 2, false>("(B) team: sc1 stores, sc1 loads", 2, 0);
    run_team<0, 1, false>("(B) team: plain stores, sc0 loads", 2, 0);
    run_team<0, 0, false>("(B) team: plain stores, plain loads (expected WRONG: L1)", 2, 0);
    run_team<3, 3, false>("(B) team: sc0 sc1 stores, sc0 sc1 loads", 3, 0);
    run_team<3, 0, false>("(B) team: sc0 sc1 stores, plain loads (expected WRONG)", 3, 0);
    run_team<3, 2, false>("(B) team: sc0 sc1 stores, sc1 loads, 5 us of work", 2, 12000);
    run_team<3, 3, false>("(B) team: sc0 sc1 stores, sc0 sc1 loads, 5 us of work", 3, 12000);
    // (C) readers on every XCD
    run_team<3, 3, true>("(C) all: sc0 sc1 stores, sc0 sc1 loads", 3, 0);
    run_team<3, 2, true>("(C) all: sc0 sc1 stores, sc1 loads (cross-XCD: WRONG?)", 3, 0);
    run_team<0, 3, true>("(C) all: plain stores, sc0 sc1 loads (WRONG?)", 3, 0);
    run_team<4, 3, true>("(C) all: nt stores, sc0 sc1 loads (WRONG?)", 3, 0);
    return 0;
}
