// tools/sync_probe.hip — how long after the last kernel does the host learn that a stream is idle?  (developer tool: the
// driver times 20 updates = 1.3 ms between two synchronisations, so ~20 us of wake-up latency are 1.5 % of the headline.)
// 80 kernels of ~15 us each, then the wait under test; reported: host time from before the first launch to after the wait,
// minus the GPU time between the first kernel's start and the last kernel's end (events).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/sync_probe.hip -o tools/sync_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin_kernel(long long ticks, float *o) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 9999) o[0] = 1.f;
}
__global__ void flag_kernel(volatile unsigned *flag, unsigned v) {
    if (threadIdx.x == 0) { __atomic_store_n((unsigned *)flag, v, __ATOMIC_RELEASE); }
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
    const unsigned sched = argc > 1 ? (unsigned)atoi(argv[1]) : 1;       // 1 spin, 2 yield, 4 blocking sync
    CK(hipSetDeviceFlags(sched == 4 ? hipDeviceScheduleBlockingSync : sched == 2 ? hipDeviceScheduleYield : hipDeviceScheduleSpin));
    hipStream_t st; CK(hipStreamCreate(&st));
    float *o; CK(hipMalloc((void **)&o, 4));
    unsigned *flag_h, *flag_d;
    CK(hipHostMalloc((void **)&flag_h, 64, hipHostMallocDefault));
    CK(hipHostGetDevicePointer((void **)&flag_d, flag_h, 0));
    *flag_h = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int NK = 80;
    const char *names[4] = {"hipStreamSynchronize", "spin on hipStreamQuery", "hipEventSynchronize on a trailing event", "flag kernel + host poll of pinned memory"};
    for (int method = 0; method < 4; ++method) {
        std::vector<double> over;
        unsigned gen = 0;
        for (int rep = 0; rep < 25; ++rep) {
            CK(hipStreamSynchronize(st));
            const double t0 = now_us();
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < NK; ++k) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, st, 1500LL, o);
            CK(hipEventRecord(e1, st));
            if (method == 0) CK(hipStreamSynchronize(st));
            else if (method == 1) { while (hipStreamQuery(st) == hipErrorNotReady) {} }
            else if (method == 2) CK(hipEventSynchronize(e1));
            else {
                ++gen;
                hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, st, flag_d, gen);
                while (__atomic_load_n(flag_h, __ATOMIC_ACQUIRE) != gen) {}
            }
            const double t1 = now_us();
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 5) over.push_back((t1 - t0) - 1e3 * ms);
        }
        std::sort(over.begin(), over.end());
        printf("schedule flag %u, %-44s host - GPU time: median %6.1f us (min %6.1f, max %6.1f)\n", sched, names[method],
               over[over.size() / 2], over.front(), over.back());
    }
    return 0;
}
