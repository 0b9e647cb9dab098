// tools/ubench_step.hip — the steady step of the tile engine in isolation (developer tool).
// One "step" = 16 DEPENDENT v_mfma_f32_16x16x4_f32 on one accumulator (the 8-wave geometry: one 16x16 tile per wave),
// the 12 fragment reads of the next chunk (8 x ds_read2_b32 + 4 x ds_read_b128), optionally a workgroup barrier.
// Two waves per SIMD (512 threads), one workgroup per CU.  Prints cycles per step for each combination, to see
// which of the costs overlap in hardware and which add.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_step.hip -o tools/ubench_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 2000;
// MF: issue the MFMAs; RD: 0 none, 1 one read behind each of MFMAs 4..15, 2 all reads before the MFMAs, 3 all reads after MFMA 1
// BAR: barrier (with lgkmcnt(0)) at the end of the step; NACC: independent accumulators the 16 MFMAs rotate over
template <bool MF, int RD, bool BAR, int NACC, int FS = 0, int FV = 0, int ND = 0, int WIN = 0, int PAT = 0>
__global__ __launch_bounds__(512) void k(float *out, long long *cyc, const char *gin = nullptr) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 24576; i += blockDim.x) sm[i] = (float)i;
    __syncthreads();
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = (float)lane, b = (float)(lane ^ 5);
    f32x2 t2[8]; f32x4 t4[4];
    for (int i = 0; i < 8; ++i) t2[i] = (f32x2){0.f, 0.f};
    for (int i = 0; i < 4; ++i) t4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // k-major P image [64][32] floats: row k = 4g + j (+16m), column l15; x-major Q image [64][64]: row l15 (+ wave), 16-byte chunk
    const int g = lane >> 4, l15 = lane & 15, w = tid >> 6;
    const uint32_t rp = (uint32_t)(uintptr_t)(sm + (4 * g) * 32 + (((l15 >> 2) ^ (4 * (g & 1))) << 2) + (l15 & 3));
    uint32_t rq[4];
    for (int m = 0; m < 4; ++m) rq[m] = (uint32_t)(uintptr_t)(sm + 8192 + ((w * 16 + l15) & 63) * 64 + ((((4 * m + g) ^ l15) & 15) << 2));
    float s = 0.f;
    unsigned sfill = 1, vfill = lane;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)sm + 65536u + (unsigned)__builtin_amdgcn_readfirstlane(w) * 1024u;
    const unsigned goffc = (unsigned)lane * 16u + (unsigned)(tid >> 6) * 4096u;
    const unsigned goffp = (unsigned)(lane >> 3) * 4224u + (unsigned)(lane & 7) * 16u + (unsigned)(tid >> 6) * 8u * 4224u;   // P: 8 rows x 128 B
    const unsigned goffq = (unsigned)(lane >> 4) * 3136u + (unsigned)(lane & 15) * 16u + (unsigned)(tid >> 6) * 4u * 3136u;  // Q: 4 rows x 256 B
    const unsigned goff = goffc;
    const char *gb0 = gin + (size_t)blockIdx.x * (WIN ? WIN : 65536);
    const char *gb = gb0;
    unsigned adv = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#define RD2(i) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(t2[i]) : "v"(rp), "n"(((i) & 3) * 2 * 32 / 1 % 256), "n"(((i) & 3) * 2 * 32 / 1 % 256 + 32))
#define RD4(i) asm volatile("ds_read_b128 %0, %1" : "=v"(t4[i]) : "v"(rq[i]))
#define MFMA(u) { if (MF) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[(u) % NACC]) : "v"(a), "v"(b)); \
                  _Pragma("unroll") for (int f_ = 0; f_ < FS; ++f_) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sfill) :: "scc"); \
                  _Pragma("unroll") for (int f_ = 0; f_ < FV; ++f_) asm volatile("v_add_u32 %0, %0, %1" : "+v"(vfill) : "v"(lane)); }
        if (RD == 2) { RD2(0); RD2(1); RD2(2); RD2(3); RD2(4); RD2(5); RD2(6); RD2(7); RD4(0); RD4(1); RD4(2); RD4(3); }
        MFMA(0);
#define DMA(n) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" :: "s"(gb + (PAT ? (n) * 32 * 3136 : (n) * 1024)), "v"(PAT ? ((n) == 0 ? goffp : goffq) : goff), "s"(lds0 + (n) * 8192u) : "memory", "m0")
        if (ND >= 1) DMA(0);
        if (ND >= 2) { MFMA(1); DMA(1); }
        if (ND >= 3) { MFMA(2); DMA(2); }
        if (ND >= 2) goto after_head;
        if (RD == 3) { RD2(0); RD2(1); RD2(2); RD2(3); RD2(4); RD2(5); RD2(6); RD2(7); RD4(0); RD4(1); RD4(2); RD4(3); }
        MFMA(1); MFMA(2);
after_head:
        if (ND == 2) MFMA(2);
        MFMA(3);
        if (RD == 1) RD2(0);
        MFMA(4);  if (RD == 1) RD2(1);
        MFMA(5);  if (RD == 1) RD2(2);
        MFMA(6);  if (RD == 1) RD2(3);
        MFMA(7);  if (RD == 1) RD2(4);
        MFMA(8);  if (RD == 1) RD2(5);
        MFMA(9);  if (RD == 1) RD2(6);
        MFMA(10); if (RD == 1) RD2(7);
        MFMA(11); if (RD == 1) RD4(0);
        MFMA(12); if (RD == 1) RD4(1);
        MFMA(13); if (RD == 1) RD4(2);
        MFMA(14); if (RD == 1) RD4(3);
        MFMA(15);
        if (ND) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ND) : "memory");
        if (WIN) { adv = (adv + 32768u) & (unsigned)(WIN - 1); gb = gb0 + adv; }
        if (BAR) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        else if (RD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1c = __builtin_readcyclecounter();
    for (int i = 0; i < 8; ++i) s += t2[i][0] + t2[i][1];
    for (int i = 0; i < 4; ++i) s += t4[i][0] + t4[i][3] + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    s += (float)sfill + (float)vfill;
    if (s == 12345.678f) out[0] = s;
    if (tid == 0) cyc[blockIdx.x] = t1c - t0;
}

template <bool MF, int RD, bool BAR, int NACC, int FS = 0, int FV = 0, int ND = 0, int WIN = 0, int PAT = 0>
static int run(const char *name, int threads, float *out, long long *cyc, const char *gin = nullptr) {
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<MF, RD, BAR, NACC, FS, FV, ND, WIN, PAT>), dim3(256), dim3(threads), 98304, 0, out, cyc, gin);
        CK(hipDeviceSynchronize());
    }
    std::vector<long long> h(256); CK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
    double c = 0; for (auto v : h) c += v; c /= 256;
    printf("%-58s %d waves/SIMD: %7.1f cycles/step\n", name, threads / 256, c / ITERS);
    return 0;
}

int main() {
    float *out; long long *cyc;
    CK(hipMalloc((void **)&out, 64)); CK(hipMalloc((void **)&cyc, 256 * 8));
#define R(MF, RD, BAR, NACC, NAME) \
    CK(hipFuncSetAttribute((const void *)k<MF, RD, BAR, NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304)); \
    if (run<MF, RD, BAR, NACC>(NAME, 512, out, cyc)) return 1; \
    if (run<MF, RD, BAR, NACC>(NAME, 256, out, cyc)) return 1;
    R(true, 0, false, 1, "16 dependent mfma")
    R(true, 0, false, 2, "16 mfma on 2 accumulators")
    R(false, 1, false, 1, "12 reads only")
    R(true, 1, false, 1, "mfma + one read behind each of mfma 4..15")
    R(true, 2, false, 1, "mfma, all reads first")
    R(true, 3, false, 1, "mfma, all reads behind mfma 1")
    R(true, 0, true, 1, "mfma + barrier")
    R(true, 1, true, 1, "mfma + interleaved reads + barrier")
    R(true, 3, true, 1, "mfma + reads behind mfma 1 + barrier")
    R(true, 1, true, 2, "2 accumulators + interleaved reads + barrier")
#define RF(FS, FV, NAME) \
    CK(hipFuncSetAttribute((const void *)k<true, 1, true, 1, FS, FV>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304)); \
    if (run<true, 1, true, 1, FS, FV>(NAME, 512, out, cyc)) return 1;
    RF(2, 0, "reads + barrier + 2 SALU per mfma")
    RF(4, 0, "reads + barrier + 4 SALU per mfma")
    RF(0, 1, "reads + barrier + 1 VALU per mfma")
    RF(0, 2, "reads + barrier + 2 VALU per mfma")
    RF(2, 1, "reads + barrier + 2 SALU + 1 VALU per mfma")
    RF(4, 2, "reads + barrier + 4 SALU + 2 VALU per mfma")
    RF(8, 2, "reads + barrier + 8 SALU + 2 VALU per mfma")
    char *gin; CK(hipMalloc((void **)&gin, (size_t)256 * 1048576 + 65536)); CK(hipMemset(gin, 0, (size_t)256 * 1048576 + 65536));
#define RD_(ND, NAME) \
    CK(hipFuncSetAttribute((const void *)k<true, 1, true, 1, 0, 0, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304)); \
    if (run<true, 1, true, 1, 0, 0, ND>(NAME, 512, out, cyc, gin)) return 1;
    RD_(1, "reads + barrier + 1 LDS-DMA per wave")
    RD_(3, "reads + barrier + 3 LDS-DMA per wave")
#define RW_(ND, WIN, NAME) \
    CK(hipFuncSetAttribute((const void *)k<true, 1, true, 1, 0, 0, ND, WIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304)); \
    if (run<true, 1, true, 1, 0, 0, ND, WIN>(NAME, 512, out, cyc, gin)) return 1;
    RW_(3, 65536, "3 LDS-DMA per wave, 64 KB window per WG (L2 hits)")
    RW_(3, 1048576, "3 LDS-DMA per wave, 1 MB window per WG (L2 misses)")
    RW_(1, 65536, "1 LDS-DMA per wave, 64 KB window per WG")
#define RP_(WIN, NAME) \
    CK(hipFuncSetAttribute((const void *)k<true, 1, true, 1, 0, 0, 3, WIN, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304)); \
    if (run<true, 1, true, 1, 0, 0, 3, WIN, 1>(NAME, 512, out, cyc, gin)) return 1;
    RP_(0, "3 LDS-DMA per wave, strided rows like the real tiles, fixed")
    RP_(262144, "3 LDS-DMA per wave, strided rows, 256 KB window per WG")
    return 0;
}
