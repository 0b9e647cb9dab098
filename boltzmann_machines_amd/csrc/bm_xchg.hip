// bm_xchg.hip — one-shot exchange over peer-mapped device memory (bm_xchg_* of include/bm355.h).
//
// SURVEY §5 ("Distributed communication backend") / §8e: the exchange step of data-parallel training is ONE
// all-reduce(sum) of the fused `grad` buffer (3.3 MB at 784 x 1024) per update, on the critical path.  A ring
// all-reduce over the point-to-point xGMI links of an 8-GPU node serialises 2 (N - 1) hops; at this size it is
// latency bound.  Here every rank maps every peer's buffer (hipIpcGetMemHandle / hipIpcOpenMemHandle: one process
// per GPU) and ONE kernel per rank does reduce-scatter + all-gather directly:
//
//   READY   rank r tells every peer "my buffer holds step e" (the producing kernel precedes this one in stream
//           order, so its writes have left the L2s), and waits for the same word from every peer;
//   reduce  rank r owns slice r (count / N floats): it reads slice r of EVERY rank's buffer - all 7 links at once,
//           16-byte system-scope loads that bypass the caches - adds them in RANK ORDER 0 .. N-1 (each element
//           is summed by exactly one rank, so all replicas receive the same bits; the order is the oracle's
//           shard algebra), and leaves the sum in its own buffer and in a staging slice `red` (write-through);
//   DONE    after a system-scope release, the last workgroup of rank r publishes "slice r is reduced";
//   gather  every rank pulls the other N - 1 reduced slices out of their owners' staging slices into its buffer.
//
// Two flag round trips and 2 (N - 1) / N of the buffer over each rank's links, in one launch.  Only LOADS cross
// the fabric (and 4-byte flag stores into fine-grained memory): what a rank's later kernels read was written by
// the rank itself, so no assumption is made about when a peer's store becomes visible to the home GPU's caches.
// The staging slice decouples the epochs: a rank overwrites `red` only after READY(e + 1) from every peer, i.e.
// after every peer has left epoch e.  Every wait is bounded (BM_XCHG_TIMEOUT_S, default 20 s): on expiry the
// kernel records an error in the status word and runs on without waiting - a lost rank shows up as an error from
// bm_xchg_status(), not as a hung GPU.
//
// bm_xchg_allreduce_max1: the same idea for ONE float (the mean-field residual of a data-parallel DBM,
// dbm.py:449-452 over the global minibatch): every rank stores {value, epoch} as one 8-byte word into every
// peer's slot and takes the max of the N words it receives - one fabric hop instead of a ring of 4-byte messages.
#include "../../include/bm355.h"
#include "bm_common.h"

#include <unistd.h>

namespace bmx {

constexpr int MAXR = 8;
constexpr int NTX = 256;
enum : int { F_READY = 0, F_DONE = MAXR, F_STATUS = 2 * MAXR, F_WORDS = 2 * MAXR + 8 };

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Blob {                       // what one rank publishes to the others (bm_xchg_export), 256 bytes
    hipIpcMemHandle_t h_buf, h_red, h_flags;       // 3 x 64 bytes
    uint64_t off_buf;                              // the registered buffer need not start its allocation
    int32_t pid, device, rank, pad;
    uint64_t count;
    char fill[256 - 3 * 64 - 8 - 16 - 8];
};
static_assert(sizeof(Blob) == 256, "blob layout");

struct Args {
    float *buf, *red;
    unsigned long long count, chunk;               // floats; chunk % 4 == 0
    int rank, n;
    unsigned epoch;
    const float *pbuf[MAXR], *pred[MAXR];          // rank r's buffer / staging slice (own entries = local pointers)
    unsigned *pflags[MAXR];                        // rank r's flag words
    unsigned *flags;                               // this rank's flag words (fine-grained memory)
    unsigned *ctr;                                 // workgroup completion counter (local)
    int grid;
    long long timeout_ticks;                       // wall_clock64 ticks (100 MHz)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, unsigned long long bytes) {
    const unsigned n = bytes > 0xfffffff0ull ? 0xfffffff0u : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)n, 0x00020000);
}
// 16-byte load that bypasses L1 / L2 (sc0 sc1): peer memory is read at its home, never from a stale line
__device__ __forceinline__ f32x4 load_sys(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 17));
}
__device__ __forceinline__ void store_sys(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)byte_off, 0, 17);
}

// wait until flags[base + r] >= epoch for every r < n (one lane per rank), bounded
__device__ __forceinline__ void wait_all(const Args &a, int base, int code) {
    const int tid = threadIdx.x;
    if (tid < a.n) {
        const long long t0 = wall_clock64();
        bool ok = true;
        while ((int)(__hip_atomic_load(a.flags + base + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - a.epoch) < 0) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > a.timeout_ticks) { ok = false; break; }
        }
        if (!ok) __hip_atomic_store(a.flags + F_STATUS, (unsigned)(code * 16 + tid + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);                 // system scope: nothing cached from before the wait
}

__global__ __launch_bounds__(NTX) void allreduce_kernel(Args a) {
    const int tid = threadIdx.x, me = a.rank, n = a.n;
    // READY(e): my buffer is complete (stream order) - tell every rank, myself included
    if (blockIdx.x == 0 && tid < n)
        __hip_atomic_store(a.pflags[tid] + F_READY + me, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    wait_all(a, F_READY, 1);
    // ---- reduce slice `me`
    const unsigned long long base = (unsigned long long)me * a.chunk;
    const unsigned long long len = base < a.count ? (a.count - base < a.chunk ? a.count - base : a.chunk) : 0ull;
    const unsigned long long len4 = (len + 3) / 4;          // float4 (the buffer's allocation is padded to 16 bytes)
    {
        __amdgpu_buffer_rsrc_t rs[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; ++r) rs[r] = rsrc(r < n ? a.pbuf[r] + base : a.buf, r < n ? len4 * 16 : 0);
        __amdgpu_buffer_rsrc_t rred = rsrc(a.red, len4 * 16);
        for (unsigned long long e = (unsigned long long)blockIdx.x * NTX + tid; e < len4; e += (unsigned long long)a.grid * NTX) {
            const unsigned off = (unsigned)(e * 16);
            f32x4 v[MAXR];
#pragma unroll
            for (int r = 0; r < MAXR; ++r) if (r < n) v[r] = load_sys(rs[r], off);
            f32x4 s = v[0];
#pragma unroll
            for (int r = 1; r < MAXR; ++r) if (r < n) s = s + v[r];          // rank order
            *reinterpret_cast<f32x4 *>(a.buf + base + e * 4) = s;
            store_sys(rred, off, s);
        }
    }
    // DONE(e): every workgroup's stores are out; the last one to arrive publishes
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == (unsigned)a.grid * a.epoch) {
            __threadfence_system();
            for (int r = 0; r < n; ++r)
                __hip_atomic_store(a.pflags[r] + F_DONE + me, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (n == 1) return;
    wait_all(a, F_DONE, 2);
    // ---- gather the other ranks' reduced slices (rotated start: the pulls spread over the links)
    for (int d = 1; d < n; ++d) {
        const int q = (me + d) % n;
        const unsigned long long bq = (unsigned long long)q * a.chunk;
        const unsigned long long lq = bq < a.count ? (a.count - bq < a.chunk ? a.count - bq : a.chunk) : 0ull;
        const unsigned long long lq4 = (lq + 3) / 4;
        __amdgpu_buffer_rsrc_t rq = rsrc(a.pred[q], lq4 * 16);
        unsigned long long e = (unsigned long long)blockIdx.x * NTX + tid;
        const unsigned long long st = (unsigned long long)a.grid * NTX;
        for (; e + st < lq4; e += 2 * st) {                  // two loads in flight per lane
            const f32x4 v0 = load_sys(rq, (unsigned)(e * 16)), v1 = load_sys(rq, (unsigned)((e + st) * 16));
            *reinterpret_cast<f32x4 *>(a.buf + bq + e * 4) = v0;
            *reinterpret_cast<f32x4 *>(a.buf + bq + (e + st) * 4) = v1;
        }
        if (e < lq4) *reinterpret_cast<f32x4 *>(a.buf + bq + e * 4) = load_sys(rq, (unsigned)(e * 16));
    }
}

// ---- all-reduce(max) of one float.  slots [2][MAXR] of {value bits, epoch} (one 8-byte store each), by epoch parity
struct MaxArgs {
    float *val;                                    // in / out (device, local)
    unsigned long long *slots;                     // this rank's slots (fine-grained)
    unsigned long long *pslots[MAXR];              // rank r's slots
    unsigned *flags;                               // status word lives here
    int rank, n;
    unsigned epoch;
    long long timeout_ticks;
};
__global__ __launch_bounds__(64) void max1_kernel(MaxArgs a) {
    const int tid = threadIdx.x;
    const unsigned par = a.epoch & 1u;
    const float mine = *a.val;
    float got = 0.f;
    if (tid < a.n) {
        const unsigned long long word = ((unsigned long long)a.epoch << 32) | (unsigned long long)__float_as_uint(mine);
        __hip_atomic_store(a.pslots[tid] + par * MAXR + a.rank, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const long long t0 = wall_clock64();
        unsigned long long w;
        bool ok = true;
        while ((unsigned)((w = __hip_atomic_load(a.slots + par * MAXR + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) >> 32) != a.epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > a.timeout_ticks) { ok = false; break; }
        }
        if (ok) got = __uint_as_float((unsigned)w);
        else __hip_atomic_store(a.flags + F_STATUS, (unsigned)(3 * 16 + tid + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) got = fmaxf(got, __shfl_xor(got, off));
    if (tid == 0) *a.val = got;                    // residuals are >= 0
}

}  // namespace bmx

struct bm_xchg {
    int rank = 0, nranks = 1, device = 0;
    float *buf = nullptr, *red = nullptr;
    size_t count = 0, chunk = 0;
    unsigned *flags = nullptr, *ctr = nullptr;     // flags: F_WORDS words + the max slots behind them
    unsigned long long *slots = nullptr;
    const float *pbuf[bmx::MAXR], *pred[bmx::MAXR];
    unsigned *pflags[bmx::MAXR];
    void *opened[3 * bmx::MAXR];
    int n_opened = 0;
    bool attached = false;
    unsigned epoch = 0, epoch_max = 0;
    int grid = 1;
    long long timeout_ticks = 0;
};

static int xchg_alloc_flags(bm_xchg *x) {
    const size_t bytes = bmx::F_WORDS * sizeof(unsigned) + 2 * bmx::MAXR * sizeof(unsigned long long);
    void *p = nullptr;
    // fine-grained, uncached device memory: written by the peers, polled here
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained); }
    BM_HIP(e);
    BM_HIP(hipMemset(p, 0, bytes));
    x->flags = (unsigned *)p;
    x->slots = (unsigned long long *)((char *)p + bmx::F_WORDS * sizeof(unsigned));
    return 0;
}

extern "C" {

int bm_xchg_create(int32_t rank, int32_t nranks, float *buf_dev, size_t count, bm_xchg **out) {
    BM_CHECK(out && buf_dev, "null argument");
    BM_CHECK(nranks >= 1 && nranks <= bmx::MAXR && rank >= 0 && rank < nranks, "bad rank %d of %d (at most %d ranks)", rank, nranks, bmx::MAXR);
    BM_CHECK(((uintptr_t)buf_dev & 15u) == 0, "the exchanged buffer must be 16-byte aligned");
    bm_xchg *x = new bm_xchg();
    x->rank = rank; x->nranks = nranks; x->buf = buf_dev; x->count = count;
    BM_HIP(hipGetDevice(&x->device));
    x->chunk = (((count + nranks - 1) / nranks) + 3) & ~(size_t)3;
    BM_HIP(hipMalloc((void **)&x->red, (x->chunk ? x->chunk : 4) * sizeof(float)));
    BM_HIP(hipMalloc((void **)&x->ctr, 64));
    BM_HIP(hipMemset(x->ctr, 0, 64));
    BM_TRY(xchg_alloc_flags(x));
    const size_t f4 = (x->chunk / 4 + bmx::NTX - 1) / bmx::NTX;
    x->grid = (int)(f4 < 1 ? 1 : (f4 > 256 ? 256 : f4));
    const char *t = getenv("BM_XCHG_TIMEOUT_S");
    x->timeout_ticks = (long long)((t ? atof(t) : 20.0) * 1e8);
    for (int r = 0; r < bmx::MAXR; ++r) { x->pbuf[r] = x->buf; x->pred[r] = x->red; x->pflags[r] = x->flags; }
    x->attached = nranks == 1;
    *out = x;
    return 0;
}

int bm_xchg_blob_bytes(void) { return (int)sizeof(bmx::Blob); }

int bm_xchg_export(bm_xchg *x, void *out_blob256) {
    BM_CHECK(x && out_blob256, "null argument");
    bmx::Blob b;
    memset(&b, 0, sizeof(b));
    void *base = nullptr; size_t sz = 0;
    BM_HIP(hipMemGetAddressRange((hipDeviceptr_t *)&base, &sz, (hipDeviceptr_t)x->buf));
    // the whole reduced range must be readable by the peers, including the 16-byte round-up of the last slice
    BM_CHECK((char *)x->buf + ((x->count + 3) & ~(size_t)3) * sizeof(float) <= (char *)base + sz,
             "the exchanged buffer's allocation must cover its length rounded up to 16 bytes");
    b.off_buf = (uint64_t)((char *)x->buf - (char *)base);
    BM_HIP(hipIpcGetMemHandle(&b.h_buf, base));
    BM_HIP(hipIpcGetMemHandle(&b.h_red, x->red));
    BM_HIP(hipIpcGetMemHandle(&b.h_flags, x->flags));
    b.pid = (int32_t)getpid(); b.device = x->device; b.rank = x->rank; b.count = x->count;
    memcpy(out_blob256, &b, sizeof(b));
    return 0;
}

int bm_xchg_attach(bm_xchg *x, const void *all_blobs) {
    BM_CHECK(x && all_blobs, "null argument");
    BM_CHECK(!x->attached || x->nranks == 1, "already attached");
    const bmx::Blob *bl = (const bmx::Blob *)all_blobs;
    for (int r = 0; r < x->nranks; ++r) {
        BM_CHECK(bl[r].rank == r, "blob %d carries rank %d: the blobs must be ordered by rank", r, bl[r].rank);
        BM_CHECK(bl[r].count == x->count, "rank %d exchanges %llu floats, this rank %zu", r, (unsigned long long)bl[r].count, x->count);
        if (r == x->rank) continue;
        BM_CHECK(bl[r].pid != (int32_t)getpid(), "rank %d lives in this process: one process per rank", r);
        void *pb = nullptr, *pr = nullptr, *pf = nullptr;
        BM_HIP(hipIpcOpenMemHandle(&pb, bl[r].h_buf, hipIpcMemLazyEnablePeerAccess));
        x->opened[x->n_opened++] = pb;
        BM_HIP(hipIpcOpenMemHandle(&pr, bl[r].h_red, hipIpcMemLazyEnablePeerAccess));
        x->opened[x->n_opened++] = pr;
        BM_HIP(hipIpcOpenMemHandle(&pf, bl[r].h_flags, hipIpcMemLazyEnablePeerAccess));
        x->opened[x->n_opened++] = pf;
        x->pbuf[r] = (const float *)((char *)pb + bl[r].off_buf);
        x->pred[r] = (const float *)pr;
        x->pflags[r] = (unsigned *)pf;
    }
    x->attached = true;
    return 0;
}

int bm_xchg_destroy(bm_xchg *x) {
    if (!x) return 0;
    (void)hipDeviceSynchronize();
    for (int i = 0; i < x->n_opened; ++i) (void)hipIpcCloseMemHandle(x->opened[i]);
    if (x->red) (void)hipFree(x->red);
    if (x->ctr) (void)hipFree(x->ctr);
    if (x->flags) (void)hipFree(x->flags);
    delete x;
    return 0;
}

// in-place all-reduce(sum) of the registered buffer, enqueued on `stream`; every rank must call it in the same order
int bm_xchg_allreduce_sum(bm_xchg *x, void *stream) {
    BM_CHECK(x && x->attached, "exchange not attached (bm_xchg_export -> gather the blobs -> bm_xchg_attach)");
    bmx::Args a;
    memset(&a, 0, sizeof(a));
    a.buf = x->buf; a.red = x->red; a.count = x->count; a.chunk = x->chunk;
    a.rank = x->rank; a.n = x->nranks; a.epoch = ++x->epoch;
    for (int r = 0; r < bmx::MAXR; ++r) { a.pbuf[r] = x->pbuf[r]; a.pred[r] = x->pred[r]; a.pflags[r] = x->pflags[r]; }
    a.flags = x->flags; a.ctr = x->ctr; a.grid = x->grid; a.timeout_ticks = x->timeout_ticks;
    hipLaunchKernelGGL(bmx::allreduce_kernel, dim3(x->grid), dim3(bmx::NTX), 0, (hipStream_t)stream, a);
    BM_HIP(hipGetLastError());
    return 0;
}

// in-place all-reduce(max) of ONE non-negative float at val_dev (device memory of this rank), enqueued on `stream`
int bm_xchg_allreduce_max1(bm_xchg *x, float *val_dev, void *stream) {
    BM_CHECK(x && x->attached && val_dev, "exchange not attached / null argument");
    bmx::MaxArgs a;
    memset(&a, 0, sizeof(a));
    a.val = val_dev; a.slots = x->slots; a.flags = x->flags; a.rank = x->rank; a.n = x->nranks;
    a.epoch = ++x->epoch_max; a.timeout_ticks = x->timeout_ticks;
    for (int r = 0; r < bmx::MAXR; ++r)
        a.pslots[r] = (unsigned long long *)((char *)x->pflags[r] + bmx::F_WORDS * sizeof(unsigned));
    hipLaunchKernelGGL(bmx::max1_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    BM_HIP(hipGetLastError());
    return 0;
}

// 0 = every wait so far was answered; otherwise 16 * phase (1 READY, 2 DONE, 3 max) + rank waited for + 1.
// Synchronises the device.
int bm_xchg_status(bm_xchg *x, int32_t *out_status) {
    BM_CHECK(x && out_status, "null argument");
    BM_HIP(hipDeviceSynchronize());
    unsigned s = 0;
    BM_HIP(hipMemcpy(&s, x->flags + bmx::F_STATUS, sizeof(s), hipMemcpyDeviceToHost));
    *out_status = (int32_t)s;
    if (s) bm::set_error("bm_xchg: wait for rank %u timed out in phase %u (1 READY, 2 DONE, 3 max)", (s & 15u) - 1u, s >> 4);
    return 0;
}

int bm_xchg_info(bm_xchg *x, int32_t *out_rank, int32_t *out_nranks, size_t *out_count) {
    BM_CHECK(x, "null argument");
    if (out_rank) *out_rank = x->rank;
    if (out_nranks) *out_nranks = x->nranks;
    if (out_count) *out_count = x->count;
    return 0;
}

// the exchange step of data-parallel training over the direct path (bm_*_allreduce_grads over RCCL is the other)
int bm_rbm_xchg_create(bm_rbm *h, int32_t rank, int32_t nranks, bm_xchg **out) {
    BM_CHECK(h && out, "null argument");
    void *p = nullptr; size_t n = 0;
    BM_TRY(bm_rbm_dev_ptr(h, "grad", &p, &n));
    return bm_xchg_create(rank, nranks, (float *)p, n, out);
}
int bm_dbm_xchg_create(bm_dbm *h, int32_t rank, int32_t nranks, bm_xchg **out) {
    BM_CHECK(h && out, "null argument");
    void *p = nullptr; size_t n = 0;
    BM_TRY(bm_dbm_dev_ptr(h, "grad", &p, &n));
    return bm_xchg_create(rank, nranks, (float *)p, n, out);
}
int bm_rbm_allreduce_grads_direct(bm_rbm *h, bm_xchg *x) {
    BM_CHECK(h && x, "null argument");
    void *p = nullptr, *st = nullptr; size_t n = 0;
    BM_TRY(bm_rbm_dev_ptr(h, "grad", &p, &n));
    BM_CHECK(p == (void *)x->buf && n == x->count, "the exchange was created for another buffer (gradient slots are not supported)");
    BM_TRY(bm_rbm_stream(h, &st));
    return bm_xchg_allreduce_sum(x, st);
}
int bm_dbm_allreduce_grads_direct(bm_dbm *h, bm_xchg *x) {
    BM_CHECK(h && x, "null argument");
    void *p = nullptr, *st = nullptr; size_t n = 0;
    BM_TRY(bm_dbm_dev_ptr(h, "grad", &p, &n));
    BM_CHECK(p == (void *)x->buf && n == x->count, "the exchange was created for another buffer");
    BM_TRY(bm_dbm_stream(h, &st));
    return bm_xchg_allreduce_sum(x, st);
}

}  // extern "C"
