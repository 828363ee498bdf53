// common.hpp - shared host-side plumbing for libsslam_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sslam_hip.h"

namespace sslam {

void set_error(const char* fmt, ...);

#define SSLAM_HIP_CHECK(expr)                                                        \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            ::sslam::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,   \
                               hipGetErrorString(_e));                               \
            return 1;                                                                \
        }                                                                            \
    } while (0)

#define SSLAM_REQUIRE(cond, ...)                                                     \
    do {                                                                             \
        if (!(cond)) {                                                               \
            ::sslam::set_error(__VA_ARGS__);                                         \
            return 2;                                                                \
        }                                                                            \
    } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Bump allocator over one hipMalloc'd slab: every instance sizes its workspace
// once at create time, so the launch path never calls hipMalloc (graph-safe).
struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0;
    bool dry = false;   // measuring pass: take() only advances `off`
    void measure() { dry = true; cap = ~(size_t)0; off = 0; base = nullptr; }
    int init(size_t bytes) {
        dry = false;
        cap = bytes;
        off = 0;
        SSLAM_HIP_CHECK(hipMalloc((void**)&base, bytes));
        SSLAM_HIP_CHECK(hipMemset(base, 0, bytes));
        return 0;
    }
    template <typename T>
    T* take(size_t n) {
        size_t o = align_up(off, 256);
        if (o + n * sizeof(T) > cap) return nullptr;
        off = o + n * sizeof(T);
        return reinterpret_cast<T*>(base + o);
    }
    void release() {
        if (base) (void)hipFree(base);
        base = nullptr;
    }
};

}  // namespace sslam

struct sslam_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // scratch for the *_host BA entry point (grown on demand, outside any graph)
    void* ba_scratch = nullptr;
    size_t ba_scratch_bytes = 0;
};
