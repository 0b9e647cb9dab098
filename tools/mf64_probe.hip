#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, double *D) {
    const int l = threadIdx.x;
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[l], B[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = acc[r];
}
int rate_main();
int main(int argc, char **argv) {
    if (argc > 1) return rate_main();
    double hA[64], hB[64], hD[256], *dA, *dB, *dD;
    for (int l = 0; l < 64; ++l) { hA[l] = 1.0 + 0.37 * l + 0.001 * l * l; hB[l] = 2.0 - 0.11 * l + 0.003 * l * l; }
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
    // candidate input layout: A lane l <-> (m = l%16, k = l/16); B lane l <-> (n = l%16, k = l/16)
    double C[16][16];
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[16 * kk + m] * hB[16 * kk + n]; C[m][n] = s; }
    int ok = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        int fm = -1, fn = -1;
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) if (fabs(C[m][n] - hD[l * 4 + r]) < 1e-9 * fabs(C[m][n])) { fm = m; fn = n; }
        if (l < 20 || l % 16 == 0) printf("lane %2d reg %d -> m=%2d n=%2d\n", l, r, fm, fn);
        ok += fm >= 0;
    }
    printf("matched %d / 256\n", ok);
    return 0;
}
// ---- rate probe: independent accumulators, no memory traffic (appended; run: ./mf64_probe rate)
__global__ __launch_bounds__(256) void rate_mfma(double *o, int iters) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
    }
    o[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
__global__ __launch_bounds__(256) void rate_fma(double *o, int iters) {
    double a[8];
    for (int q = 0; q < 8; ++q) a[q] = q;
    double x = 1.0 + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = fma(a[q], x, 1e-9);
    double s = 0; for (int q = 0; q < 8; ++q) s += a[q];
    o[blockIdx.x * 256 + threadIdx.x] = s;
}
int rate_main() {
    double *o; hipMalloc(&o, 1024 * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(rate_mfma, dim3(1024), dim3(256), 0, 0, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mfma_f64_16x16x4: %.1f TFLOP/s\n", 1024.0 * 4 * iters * 4 * 2048.0 / (ms * 1e-3) / 1e12);
        hipEventRecord(e0); hipLaunchKernelGGL(rate_fma, dim3(1024), dim3(256), 0, 0, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("v_fma_f64       : %.1f TFLOP/s\n", 1024.0 * 256 * iters * 8 * 2.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
