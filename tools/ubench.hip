// tools/ubench.hip — issue-model microbenchmarks for gfx950 (developer tool).
// How many cycles does one wave (or two per SIMD) need for a stream of v_mfma_f32_16x16x4_f32
// interleaved with VALU / LDS / VMEM instructions?  Prints cycles per MFMA.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 2000;
// MODE: 0 mfma only; 1 +NX v_fma per mfma; 2 +NX v_mul_hi per mfma; 3 +ds_read_b64 every NX mfma;
//       4 +ds_write_b128 every NX mfma; 5 +global_load_dwordx4 every NX mfma; 6 ds_read_b64 only; 7 ds_write_b128 only
//       8 ds_read_b32 every NX mfma; 9: v_fma only (NX per slot)
template <int MODE, int NX, bool MF>
__global__ __launch_bounds__(512) void k(float *out, const float4 *gin, long long *cyc) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += blockDim.x) sm[i] = (float)i;
    __syncthreads();
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float va[8]; uint32_t ua[8];
    for (int i = 0; i < 8; ++i) { va[i] = (float)(tid + i); ua[i] = tid * 7 + i; }
    float a = (float)lane, b = (float)(lane ^ 5), c1 = 1.0001f, c2 = 0.5f;
    uint32_t mm = 0xD2511F53u;
    f32x2 t2[16]; float t1[16]; f32x4 t4[16];
    for (int i = 0; i < 16; ++i) { t2[i] = (f32x2){0.f, 0.f}; t1[i] = 0.f; t4[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    f32x4 w4 = (f32x4){1.f, 2.f, 3.f, 4.f};
    const uint32_t rp = (uint32_t)(uintptr_t)(sm + ((lane >> 4) * 96 + 2 * (lane & 15)));     // conflict-free b64 pattern
    const uint32_t rp1 = (uint32_t)(uintptr_t)(sm + ((lane >> 4) * 80 + (lane & 15)));        // conflict-free b32 pattern
    const uint32_t wp = (uint32_t)(uintptr_t)(sm + 8192 + (tid & 255) * 4);
    const float4 *gp = gin + tid;
    float s = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MF) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
            if (MODE == 1 || MODE == 9) {
#pragma unroll
                for (int x = 0; x < NX; ++x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(va[(u + x) & 7]) : "v"(c1), "v"(c2));
            }
            if (MODE == 2) {
#pragma unroll
                for (int x = 0; x < NX; ++x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(ua[(u + x) & 7]) : "v"(mm));
            }
            if ((MODE == 3 || MODE == 6) && u % NX == 0) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t2[u]) : "v"(rp), "n"((u & 7) * 4 * 96 * 4));
            if (MODE == 8 && u % NX == 0) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t1[u]) : "v"(rp1), "n"((u & 7) * 4 * 80 * 4));
            if ((MODE == 4 || MODE == 7) && u % NX == 0) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(wp), "v"(w4), "n"((u & 3) * 4096));
            if (MODE == 5 && u % NX == 0) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(t4[u]) : "v"(gp), "n"((u & 7) * 512));
        }
        if (MODE == 3 || MODE == 6 || MODE == 8 || MODE == 4 || MODE == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long t1c = __builtin_readcyclecounter();
    for (int i = 0; i < 16; ++i) s += t2[i][0] + t2[i][1] + t1[i] + t4[i][0] + t4[i][3];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += va[i] + (float)ua[i];
    if (s == 12345.678f) out[0] = s;
    if (tid == 0) cyc[blockIdx.x] = t1c - t0;
}

template <int MODE, int NX, bool MF>
static int run(const char *name, int threads, float *out, float4 *gin, long long *cyc) {
    hipLaunchKernelGGL((k<MODE, NX, MF>), dim3(256), dim3(threads), 65536, 0, out, gin, cyc);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<MODE, NX, MF>), dim3(256), dim3(threads), 65536, 0, out, gin, cyc);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(256); CK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
    double c = 0; for (auto v : h) c += v; c /= 256;
    const double slots = (double)ITERS * 16;
    printf("%-44s %d waves/SIMD: %7.2f cycles/slot  (%.1f us, counter %.3f GHz)\n", name, threads / 256, c / slots, ms * 1e3, c / (ms * 1e6));
    return 0;
}
#define R(MODE, NX, MF, NAME) run<MODE, NX, MF>(NAME, 256, out, gin, cyc); run<MODE, NX, MF>(NAME, 512, out, gin, cyc);
int main() {
    float *out; float4 *gin; long long *cyc;
    CK(hipMalloc((void **)&out, 64)); CK(hipMalloc((void **)&gin, 8192 * 16)); CK(hipMalloc((void **)&cyc, 256 * 8));
    CK(hipMemset(gin, 0, 8192 * 16));
    R(0, 1, true, "mfma only")
    R(1, 1, true, "mfma + 1 v_fma")
    R(1, 2, true, "mfma + 2 v_fma")
    R(1, 4, true, "mfma + 4 v_fma")
    R(1, 6, true, "mfma + 6 v_fma")
    R(1, 8, true, "mfma + 8 v_fma")
    R(9, 4, false, "4 v_fma only")
    R(2, 1, true, "mfma + 1 v_mul_hi")
    R(2, 2, true, "mfma + 2 v_mul_hi")
    R(2, 2, false, "2 v_mul_hi only")
    R(3, 1, true, "mfma + 1 ds_read_b64")
    R(3, 2, true, "mfma + 1/2 ds_read_b64")
    R(6, 1, false, "ds_read_b64 only")
    R(8, 1, true, "mfma + 1 ds_read_b32")
    R(8, 1, false, "ds_read_b32 only")
    R(4, 2, true, "mfma + 1/2 ds_write_b128")
    R(4, 4, true, "mfma + 1/4 ds_write_b128")
    R(7, 1, false, "ds_write_b128 only")
    R(5, 4, true, "mfma + 1/4 global_load_dwordx4")
    R(5, 2, true, "mfma + 1/2 global_load_dwordx4")
    R(5, 1, false, "global_load_dwordx4 only")
    return 0;
}
