// bm_common.h — error plumbing + device buffer helper shared by the API files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <map>
#include <vector>

namespace bm {

void set_error(const char *fmt, ...);

// Developer switches live in ONE environment variable: BM355_DEBUG="name=value,name=value" (DESIGN.md 9 lists the names:
// forced tile geometries, the launch tuner's log, chained-launch modes and measurement aids).  dbg("name") returns the value
// text or null.  None of them changes results; the variables a USER may set (BM355_HOST_WAIT, BM355_FAST_BINARY,
// BM355_AIS_LITERAL, BM355_DATA_PARALLEL, BM355_STAGED_SAVE, BM355_RCCL_LIB, BM_XCHG_TIMEOUT_S) keep names of their own.
static inline const char *dbg(const char *name) {
    static const std::map<std::string, std::string> *tab = [] {
        auto *m = new std::map<std::string, std::string>();
        const char *e = getenv("BM355_DEBUG");
        std::string cur;
        for (const char *c = e ? e : ""; ; ++c) {
            if (*c == ',' || *c == ';' || *c == ' ' || *c == 0) {
                if (!cur.empty()) {
                    const size_t q = cur.find('=');
                    (*m)[q == std::string::npos ? cur : cur.substr(0, q)] = q == std::string::npos ? std::string("1") : cur.substr(q + 1);
                }
                cur.clear();
                if (*c == 0) break;
            } else cur += *c;
        }
        return m;
    }();
    const auto it = tab->find(name);
    return it == tab->end() ? nullptr : it->second.c_str();
}

#define BM_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            bm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                          __LINE__);                                                          \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

#define BM_CHECK(cond, ...)                 \
    do {                                    \
        if (!(cond)) {                      \
            bm::set_error(__VA_ARGS__);     \
            return 2;                       \
        }                                   \
    } while (0)

#define BM_TRY(expr)              \
    do {                          \
        int _r = (expr);          \
        if (_r) return _r;        \
    } while (0)

struct DevBuf {
    float *p = nullptr;
    size_t n = 0;
    int alloc(size_t count) {
        n = count;
        count = (count + 3) & ~(size_t)3;      // whole 16-byte groups: vector kernels (bm_xchg) may touch the round-up
        if (count == 0) count = 4;
        BM_HIP(hipMalloc((void **)&p, count * sizeof(float)));
        BM_HIP(hipMemset(p, 0, count * sizeof(float)));
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

// Leading dimension for matrices owned by the library.  A row pitch that is a
// multiple of 256 B (e.g. H = 1024 floats = 4 KiB) makes every row of a K-chunk
// hit the SAME L2/HBM channel (measured: 6x slower tiles); such pitches get +128 B.
static inline int pad_ld(int n) {
    int ld = (n + 3) & ~3;
    if (ld % 64 == 0) ld += 32;
    return ld;
}

// row-major [rows][cols] matrix in HBM with padded pitch `ld`
struct Mat {
    float *p = nullptr;
    int rows = 0, cols = 0, ld = 0;
    size_t count() const { return (size_t)rows * ld; }
    int alloc(int r, int c) {
        rows = r; cols = c; ld = pad_ld(c);
        size_t cnt = count() ? count() : 1;
        BM_HIP(hipMalloc((void **)&p, cnt * sizeof(float)));
        BM_HIP(hipMemset(p, 0, cnt * sizeof(float)));
        return 0;
    }
    int upload(const float *host) {   // dense host [rows][cols] -> device
        BM_HIP(hipMemcpy2D(p, (size_t)ld * sizeof(float), host, (size_t)cols * sizeof(float),
                           (size_t)cols * sizeof(float), rows, hipMemcpyHostToDevice));
        return 0;
    }
    int download(float *host) const {
        BM_HIP(hipMemcpy2D(host, (size_t)cols * sizeof(float), p, (size_t)ld * sizeof(float),
                           (size_t)cols * sizeof(float), rows, hipMemcpyDeviceToHost));
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
};

// bf16 matrix [planes][rows][ld] (bm_bf3.h): weight planes or the shadow of a {0,1} state matrix; ld % 64 == 0,
// zero initialised (the padding must stay zero: the bf16 contraction has no K tail handling)
struct Mat16 {
    uint16_t *p = nullptr;
    int planes = 0, rows = 0, cols = 0, ld = 0;
    long long plane_stride() const { return (long long)rows * ld; }
    int alloc(int np, int r, int c) {
        release();
        planes = np; rows = r; cols = c; ld = (c + 63) & ~63;
        const size_t bytes = (size_t)np * r * ld * sizeof(uint16_t);
        BM_HIP(hipMalloc((void **)&p, bytes ? bytes : 2));
        BM_HIP(hipMemset(p, 0, bytes ? bytes : 2));
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; planes = rows = cols = ld = 0;
    }
};

}  // namespace bm
