// tools/chainprobe.hip — what does a producer -> consumer hand-over between WORKGROUPS OF ONE LAUNCH cost, against a kernel
// boundary?  (developer tool for the "dependent workgroups instead of dependent launches" item of DESIGN.md §8.)
//
// P phases of NB workgroups each; a workgroup of phase p reads 8 KB that a workgroup of phase p-1 ON ANOTHER XCD wrote
// (ping-pong over two buffers, so stale lines of phase p-2 sit in every L2), checks the values, spins `work` cycles,
// writes its own 8 KB, and signals the per-group counter its readers poll (groups of 32 producers = one row block of the
// tile engine).  Run (a) as P launches, (b) as ONE launch of P*NB workgroups with the agent-scope fences of the LLVM memory
// model (release = L2 writeback, acquire = L2 invalidate), (c) as one launch with write-through stores / L2-bypassing loads
// (sc0 sc1) and no L2 maintenance.  Spins are bounded: the kernel cannot hang.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/chainprobe.hip -o tools/chainprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int NB = 256, WORDS = 2048, GROUP = 32, NG = NB / GROUP;

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE> __device__ __forceinline__ f32x4 ld4(const float *p) {
    f32x4 v;
    if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else v = *reinterpret_cast<const f32x4 *>(p);
    return v;
}
template <int MODE> __device__ __forceinline__ void st4(float *p, f32x4 v) {
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p));
}

// MODE 0: separate launches (phase = argument, no waits)   1: chained, agent fences   2: chained, sc0 sc1 accesses
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned *cnt, float *buf0, float *buf1, int phase_arg, int nphases, int work, unsigned gen,
                                         int *err, long long *stamps) {
    __shared__ float big[30 * 1024];                 // 120 KiB: one workgroup per CU, as the tile kernels
    const int p = MODE == 0 ? phase_arg : (int)blockIdx.x / NB, b = (int)blockIdx.x % NB;
    const long long t0 = wall_clock64();
    const float *src = (p & 1) ? buf0 : buf1;
    float *dst = (p & 1) ? buf1 : buf0;
    const int o = (b * 37 + 11) % NB;                 // producer on (almost always) another XCD
    int bad = 0;
    long long t1 = t0;
    if (p > 0) {
        if (MODE != 0) {
            if (threadIdx.x == 0) {
                const unsigned *c = cnt + ((size_t)(p - 1) * NG + o / GROUP) * 32;   // 128-byte spacing
                const unsigned target = gen * GROUP;
                unsigned spins = 0;
                while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spins > (1u << 20)) { atomicExch(err, 1); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        }
        t1 = wall_clock64();
        const float expect = (float)((p - 1) * 1024 + o) + (float)gen;
        for (int i = threadIdx.x * 4; i < WORDS; i += blockDim.x * 4) {
            const f32x4 v = ld4<MODE>(src + (size_t)o * WORDS + i);
            bad += (v[0] != expect) + (v[1] != expect) + (v[2] != expect) + (v[3] != expect);
        }
    }
    const long long t2 = wall_clock64();
    // "work": a dependent chain on the cycle counter
    const long long c0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - c0 < work) big[threadIdx.x] += 1.f;
    const float val = (float)(p * 1024 + b) + (float)gen;
    for (int i = threadIdx.x * 4; i < WORDS; i += blockDim.x * 4) st4<MODE>(dst + (size_t)b * WORDS + i, (f32x4){val, val, val, val});
    if (MODE != 0 && p + 1 < nphases) {
        if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_fetch_add(cnt + ((size_t)p * NG + b / GROUP) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (bad) atomicAdd(err + 1, bad);
    if (threadIdx.x == 0 && stamps) {
        long long *s = stamps + ((size_t)p * NB + b) * 4;
        s[0] = t0; s[1] = t1; s[2] = t2; s[3] = wall_clock64();
    }
}

int main(int argc, char **argv) {
    const int P = 8, REP = 50;
    const int work = argc > 1 ? atoi(argv[1]) : 12000;          // cycles of "work" per workgroup (~5 us at 2.4 GHz)
    unsigned *cnt; float *buf0, *buf1; int *err; long long *stamps;
    CK(hipMalloc((void **)&cnt, (size_t)P * NG * 32 * 4)); CK(hipMalloc((void **)&buf0, (size_t)NB * WORDS * 4));
    CK(hipMalloc((void **)&buf1, (size_t)NB * WORDS * 4)); CK(hipMalloc((void **)&err, 8));
    CK(hipMalloc((void **)&stamps, (size_t)P * NB * 4 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipMemset(cnt, 0, (size_t)P * NG * 32 * 4)); CK(hipMemset(err, 0, 8));
        float best = 1e9f, sum = 0.f;
        for (int rep = 0; rep < REP + 5; ++rep) {
            const unsigned gen = rep + 1;
            CK(hipEventRecord(e0, 0));
            if (mode == 0) for (int p = 0; p < P; ++p) hipLaunchKernelGGL(k<0>, dim3(NB), dim3(512), 0, 0, cnt, buf0, buf1, p, P, work, gen, err, stamps);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(NB * P), dim3(512), 0, 0, cnt, buf0, buf1, 0, P, work, gen, err, stamps);
            else hipLaunchKernelGGL(k<2>, dim3(NB * P), dim3(512), 0, 0, cnt, buf0, buf1, 0, P, work, gen, err, stamps);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 5) { sum += ms; if (ms < best) best = ms; }
        }
        int herr[2]; CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
        std::vector<long long> st((size_t)P * NB * 4); CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
        // per phase boundary: last producer end -> median consumer "data may be read" (t1), in 10 ns ticks of the 100 MHz clock
        double gap = 0, rd = 0, wait_in = 0;
        for (int p = 1; p < P; ++p) {
            long long last_end = 0;
            for (int b = 0; b < NB; ++b) last_end = st[((size_t)(p - 1) * NB + b) * 4 + 3] > last_end ? st[((size_t)(p - 1) * NB + b) * 4 + 3] : last_end;
            std::vector<long long> t1s, rds, w;
            for (int b = 0; b < NB; ++b) {
                const long long *s = &st[((size_t)p * NB + b) * 4];
                t1s.push_back(s[1] - last_end); rds.push_back(s[2] - s[1]); w.push_back(s[1] - s[0]);
            }
            std::sort(t1s.begin(), t1s.end()); std::sort(rds.begin(), rds.end()); std::sort(w.begin(), w.end());
            gap += t1s[NB / 2]; rd += rds[NB / 2]; wait_in += w[NB / 2];
        }
        printf("%s: %d phases of %d workgroups, work %d cycles: %.1f us per phase (best %.1f); last producer end -> median consumer ready %.2f us, "
               "median time inside the wait %.2f us, median 8 KB read %.2f us; timeouts %d, wrong values %d\n",
               mode == 0 ? "separate launches        " : mode == 1 ? "one launch, agent fences " : "one launch, sc0 sc1      ", P, NB, work,
               sum / REP * 1e3 / P, best * 1e3 / P, gap / (P - 1) * 0.01, wait_in / (P - 1) * 0.01, rd / (P - 1) * 0.01, herr[0], herr[1]);
    }
    return 0;
}
