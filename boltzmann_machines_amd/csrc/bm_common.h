// bm_common.h — error plumbing + device buffer helper shared by the API files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <map>
#include <vector>

namespace bm {

void set_error(const char *fmt, ...);

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
        if (count == 0) count = 1;
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

}  // namespace bm
