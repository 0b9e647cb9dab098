// bm_comm.hip — RCCL inside the library (bm_comm_* / bm_*_allreduce_grads of include/bm355.h).
//
// one exchange step per update — an all-reduce(sum) of the fused `grad` buffer,
// enqueued on the handle's own HIP stream between bm_*_grad_step and bm_*_apply_step.  The host
// only carries the 128-byte ncclUniqueId from rank 0 to the other ranks (any channel: MPI, a
// file, torch.distributed's store) and calls bm_comm_init on every rank.
// librccl is opened with dlopen at the first use: libbm355.so itself does not link it, so the
// single-GPU path has no dependency on it and a host that brings its own RCCL (PyTorch bundles
// one) is not forced to share symbols with ours.
#include <dlfcn.h>
#include "bm_common.h"

namespace bmcomm {
typedef struct { char internal[128]; } UniqueId;       // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void *Comm;                                     // ncclComm_t
typedef int Result;                                     // ncclResult_t, 0 = ncclSuccess
enum { kSum = 0, kMax = 2, kFloat32 = 7 };              // ncclSum, ncclMax, ncclFloat32 (rccl.h)
struct Api {
    void *lib = nullptr;
    Result (*GetUniqueId)(UniqueId *) = nullptr;
    Result (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    Result (*CommDestroy)(Comm) = nullptr;
    Result (*AllReduce)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    Result (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(Result) = nullptr;
};
static Api g_api;
static int load_api() {
    if (g_api.lib) return 0;
    const char *names[] = {getenv("BM355_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names) {
        if (!n || !*n) continue;
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
    }
    BM_CHECK(lib, "cannot open librccl (set BM355_RCCL_LIB): %s", dlerror());
    Api a;
    a.lib = lib;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(lib, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(lib, "ncclAllReduce");
    a.AllGather = (decltype(a.AllGather))dlsym(lib, "ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
    BM_CHECK(a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.AllGather, "librccl lacks a required symbol");
    g_api = a;
    return 0;
}
#define BM_NCCL(call)                                                                               \
    do {                                                                                            \
        const bmcomm::Result r_ = (call);                                                           \
        if (r_ != 0) {                                                                              \
            bm::set_error("%s failed: %s", #call, bmcomm::g_api.GetErrorString ? bmcomm::g_api.GetErrorString(r_) : "rccl error"); \
            return 1;                                                                               \
        }                                                                                           \
    } while (0)
}  // namespace bmcomm

struct bm_comm {
    bmcomm::Comm comm = nullptr;
    int rank = 0, nranks = 1;
};

extern "C" {

int bm_comm_unique_id(void *out_id128) {
    BM_CHECK(out_id128, "null argument");
    BM_TRY(bmcomm::load_api());
    bmcomm::UniqueId id;
    BM_NCCL(bmcomm::g_api.GetUniqueId(&id));
    memcpy(out_id128, &id, sizeof(id));
    return 0;
}

int bm_comm_init(int32_t rank, int32_t nranks, const void *id128, bm_comm **out) {
    BM_CHECK(out && id128, "null argument");
    BM_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
    BM_TRY(bmcomm::load_api());
    bmcomm::UniqueId id;
    memcpy(&id, id128, sizeof(id));
    bm_comm *c = new bm_comm();
    c->rank = rank; c->nranks = nranks;
    const bmcomm::Result r = bmcomm::g_api.CommInitRank(&c->comm, nranks, id, rank);     // on the CURRENT device
    if (r != 0) {
        bm::set_error("ncclCommInitRank failed: %s", bmcomm::g_api.GetErrorString ? bmcomm::g_api.GetErrorString(r) : "rccl error");
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}

int bm_comm_destroy(bm_comm *c) {
    if (!c) return 0;
    if (c->comm) (void)bmcomm::g_api.CommDestroy(c->comm);
    delete c;
    return 0;
}

// in-place all-reduce(sum) of `count` floats at `buf_dev`, enqueued on `stream` (a hipStream_t as void*)
int bm_comm_allreduce_sum(bm_comm *c, float *buf_dev, size_t count, void *stream) {
    BM_CHECK(c && c->comm && buf_dev, "null argument");
    BM_NCCL(bmcomm::g_api.AllReduce(buf_dev, buf_dev, count, bmcomm::kFloat32, bmcomm::kSum, c->comm, (hipStream_t)stream));
    return 0;
}

// in-place all-reduce(max) (the mean-field residual of a data-parallel DBM: one float per sweep)
int bm_comm_allreduce_max(bm_comm *c, float *buf_dev, size_t count, void *stream) {
    BM_CHECK(c && c->comm && buf_dev, "null argument");
    BM_NCCL(bmcomm::g_api.AllReduce(buf_dev, buf_dev, count, bmcomm::kFloat32, bmcomm::kMax, c->comm, (hipStream_t)stream));
    return 0;
}

int bm_comm_rank(bm_comm *c, int32_t *out_rank, int32_t *out_nranks) {
    BM_CHECK(c, "null argument");
    if (out_rank) *out_rank = c->rank;
    if (out_nranks) *out_nranks = c->nranks;
    return 0;
}

// all-gather of `count` floats per rank (AIS log-weights, equal chain counts per rank)
int bm_comm_allgather(bm_comm *c, const float *send_dev, float *recv_dev, size_t count, void *stream) {
    BM_CHECK(c && c->comm && send_dev && recv_dev, "null argument");
    BM_NCCL(bmcomm::g_api.AllGather(send_dev, recv_dev, count, bmcomm::kFloat32, c->comm, (hipStream_t)stream));
    return 0;
}

int bm_rbm_allreduce_grads(bm_rbm *h, bm_comm *c) {
    BM_CHECK(h && c, "null argument");
    void *p = nullptr, *st = nullptr;
    size_t n = 0;
    BM_TRY(bm_rbm_dev_ptr(h, "grad", &p, &n));
    BM_TRY(bm_rbm_stream(h, &st));
    return bm_comm_allreduce_sum(c, (float *)p, n, st);
}

int bm_dbm_allreduce_grads(bm_dbm *h, bm_comm *c) {
    BM_CHECK(h && c, "null argument");
    void *p = nullptr, *st = nullptr;
    size_t n = 0;
    BM_TRY(bm_dbm_dev_ptr(h, "grad", &p, &n));
    BM_TRY(bm_dbm_stream(h, &st));
    return bm_comm_allreduce_sum(c, (float *)p, n, st);
}

}  // extern "C"
