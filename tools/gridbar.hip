// tools/gridbar.hip — cost and correctness of a hand-rolled grid barrier across the 8 XCDs
// (developer tool: is a persistent CD-k kernel worth building?).  Spins are bounded, the kernel
// cannot hang.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gridbar.hip -o tools/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned nblocks, unsigned &phase, int *err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();                                   // release: L2 writeback (agent scope)
        const unsigned target = (phase + 1) * nblocks;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1u << 22)) { ok = false; atomicExch(err, 1); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();                                   // acquire: L2 invalidate
    }
    __syncthreads();
    ++phase;
    return ok;
}

// every phase: each block writes `words` floats (its slice), barrier, reads the slice of a block on another XCD
__global__ __launch_bounds__(512) void k(unsigned *counter, float *buf, int words, int phases, int *err, long long *cyc, int do_data) {
    unsigned phase = 0;
    const int b = blockIdx.x, nb = gridDim.x;
    float *mine = buf + (size_t)b * words;
    int bad = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int p = 0; p < phases; ++p) {
        if (do_data) for (int i = threadIdx.x; i < words; i += blockDim.x) mine[i] = (float)(p * 1024 + b);
        if (!grid_barrier(counter, nb, phase, err)) break;
        if (do_data) {
            const int o = (b + 37) % nb;
            const float *other = buf + (size_t)o * words;
            for (int i = threadIdx.x; i < words; i += blockDim.x) if (other[i] != (float)(p * 1024 + o)) ++bad;
        }
        if (!grid_barrier(counter, nb, phase, err)) break;     // nobody overwrites before all have read
    }
    const long long t1 = __builtin_readcyclecounter();
    if (bad) atomicAdd(err + 1, bad);
    if (threadIdx.x == 0) cyc[b] = t1 - t0;
}

// group barrier: 8 groups (blockIdx % 8 = the XCD under round-robin dispatch) of nb/8 workgroups, one counter each.
// FENCE 2: agent-scope fences (L2 writeback + invalidate); 1: no L2 maintenance, only the L1/K-cache acquire that
// a same-XCD consumer needs (valid only if a group really sits on one XCD); the XCC_ID register is recorded.
template <int FENCE>
__global__ __launch_bounds__(512) void kg(unsigned *counters, float *buf, int words, int phases, int *err, int *xcc) {
    unsigned phase = 0;
    const int b = blockIdx.x, nb = gridDim.x, grp = b % 8, ngrp = nb / 8;
    unsigned *counter = counters + grp * 64;
    float *mine = buf + (size_t)b * words;
    int bad = 0;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[b] = (int)(id & 0xf);
    }
    for (int p = 0; p < phases; ++p) {
        for (int i = threadIdx.x; i < words; i += blockDim.x) mine[i] = (float)(p * 1024 + b);
        for (int half = 0; half < 2; ++half) {
            if (FENCE != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this thread's stores are in the L2
            __syncthreads();
            if (threadIdx.x == 0) {
                if (FENCE == 2) __threadfence();
                const unsigned target = (phase + 1) * ngrp;
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spins > (1u << 22)) { atomicExch(err, 1); break; }
                }
                if (FENCE == 2) __threadfence();
                if (FENCE == 1) asm volatile("buffer_inv sc1\n s_waitcnt vmcnt(0)" ::: "memory");   // ONE wave per CU invalidates
            }
            __syncthreads();
            ++phase;
            if (half == 0) {
                const int o = ((b / 8 + 5) % ngrp) * 8 + grp;      // another workgroup of the SAME group
                const float *other = buf + (size_t)o * words;
                for (int i = threadIdx.x; i < words; i += blockDim.x) if (other[i] != (float)(p * 1024 + o)) ++bad;
            }
        }
    }
    if (bad) atomicAdd(err + 1, bad);
}

int main() {
    unsigned *counter; float *buf; int *err; long long *cyc;
    const int NB = 256, WORDS = 8192, PH = 500;       // 32 KB per block per phase = 8 MB per phase
    CK(hipMalloc((void **)&counter, 4)); CK(hipMalloc((void **)&buf, (size_t)NB * WORDS * 4));
    CK(hipMalloc((void **)&err, 8)); CK(hipMalloc((void **)&cyc, NB * 8));
    for (int data = 0; data < 2; ++data) {
        CK(hipMemset(counter, 0, 4)); CK(hipMemset(err, 0, 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(NB), dim3(512), 0, 0, counter, buf, WORDS, PH, err, cyc, data);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int herr[2]; CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
        printf("%s: %d phases x 2 barriers in %.1f us = %.2f us per barrier%s; timeout %d, wrong reads %d\n",
               data ? "with 8 MB written+read per phase" : "barrier only", PH, ms * 1e3, ms * 1e3 / (2 * PH),
               data ? " (incl. the data movement)" : "", herr[0], herr[1]);
    }
    {   // group barriers
        unsigned *counters; int *xcc;
        CK(hipMalloc((void **)&counters, 8 * 64 * 4)); CK(hipMalloc((void **)&xcc, NB * 4));
        for (int fence = 2; fence >= 0; --fence) {
            CK(hipMemset(counters, 0, 8 * 64 * 4)); CK(hipMemset(err, 0, 8));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            if (fence == 2) hipLaunchKernelGGL(kg<2>, dim3(NB), dim3(512), 0, 0, counters, buf, 2048, PH, err, xcc);
            else if (fence == 1) hipLaunchKernelGGL(kg<1>, dim3(NB), dim3(512), 0, 0, counters, buf, 2048, PH, err, xcc);
            else            hipLaunchKernelGGL(kg<0>, dim3(NB), dim3(512), 0, 0, counters, buf, 2048, PH, err, xcc);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int herr[2]; CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
            std::vector<int> hx(NB); CK(hipMemcpy(hx.data(), xcc, NB * 4, hipMemcpyDeviceToHost));
            int mism = 0; for (int b = 0; b < NB; ++b) if (hx[b] != hx[b % 8]) ++mism;
            printf("group barrier (8 groups of %d, 8 KB written+read per wg and phase), %s: %.2f us per barrier; timeout %d, wrong reads %d; "
                   "workgroups not on the XCD of their group leader: %d (XCC ids of blocks 0..7: %d %d %d %d %d %d %d %d)\n", NB / 8,
                   fence == 2 ? "agent fences" : fence == 1 ? "vmcnt(0) + one buffer_inv sc1 per workgroup" : "vmcnt(0) only", ms * 1e3 / (2 * PH), herr[0], herr[1], mism,
                   hx[0], hx[1], hx[2], hx[3], hx[4], hx[5], hx[6], hx[7]);
        }
    }
    return 0;
}
