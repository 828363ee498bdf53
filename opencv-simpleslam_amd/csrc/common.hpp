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

// Launch-sequence cache.  The "_dev" entry points enqueue fixed kernel sequences whose control flow
// lives on the device, so for a given set of arguments (pointers, sizes, instance settings) the
// whole sequence is one hipGraph: the first call with a key captures the stream into a graph,
// later calls replay it with one hipGraphLaunch (~10 us of host time instead of ~2.6 us per
// launch for 45 - 190 launches).  Bounded LRU; cleared whenever an instance setting changes.
// The key holds every pointer argument: a caller that passes FRESH buffers on every call never
// hits (capture + instantiate per call, slower than plain launches) - graphs are for callers that
// cycle through a fixed set of buffers, as the frame pipeline does.
//
// SEGMENTS (r06).  A graph of short kernels does not replay as one submission on this runtime: rocprofv3 shows the chain
// stop for 25 - 30 us after its 15th kernel node (and again after the 31st and the 63rd) while plain launches of the same
// sequence run back to back (profiles/r06_graph_segments.md) - the executor hands the device the nodes in batches and the
// next batch only when the previous one has finished.  A sequence may therefore mark cut points (`graph_cut(s)` inside the
// enqueue function, at most ~12 launches apart): each piece is captured and instantiated as a graph of its own and a replay
// is one hipGraphLaunch per piece, all of them queued at once - no mid-sequence stop, still ~10 us of host time per piece
// instead of ~2.6 us per launch.
struct GraphCapture { std::vector<hipGraph_t> graphs; std::vector<hipGraphExec_t> execs; hipStream_t s = nullptr; bool failed = false; };
inline thread_local GraphCapture* t_capture = nullptr;

inline bool capture_end_segment_(GraphCapture& c) {           // the piece captured so far -> an executable graph
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(c.s, &g);
    if (e != hipSuccess || !g) { set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); c.failed = true; return false; }
    hipGraphExec_t x = nullptr;
    if (const hipError_t ei = hipGraphInstantiate(&x, g, nullptr, nullptr, 0); ei != hipSuccess) {
        (void)hipGraphDestroy(g);
        set_error("hipGraphInstantiate: %s", hipGetErrorString(ei));
        c.failed = true;
        return false;
    }
    c.graphs.push_back(g); c.execs.push_back(x);
    return true;
}

// Inside an enqueue function: the launches so far and the launches that follow replay as separate graphs.  A no-op
// outside a capture (plain launches) and for another stream's capture.
inline void graph_cut(hipStream_t s) {
    GraphCapture* c = t_capture;
    if (!c || c->s != s || c->failed) return;
    if (!capture_end_segment_(*c)) return;
    if (const hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal); e != hipSuccess) {
        set_error("hipStreamBeginCapture: %s", hipGetErrorString(e));
        c->failed = true;
    }
}

struct GraphCache {
    struct Entry { std::vector<uint64_t> key; std::vector<hipGraph_t> graphs; std::vector<hipGraphExec_t> execs; uint64_t stamp; };
    std::vector<Entry> entries;
    uint64_t clock = 0;
    size_t cap = 128;
    uint64_t hits = 0, misses = 0;
    const std::vector<hipGraphExec_t>* find(const std::vector<uint64_t>& key) {
        for (auto& e : entries)
            if (e.key == key) { e.stamp = ++clock; ++hits; return &e.execs; }
        ++misses;
        return nullptr;
    }
    static void destroy_(Entry& e) {
        for (auto x : e.execs) (void)hipGraphExecDestroy(x);
        for (auto g : e.graphs) (void)hipGraphDestroy(g);
    }
    // `s`: the stream the cached graphs are launched on - an evicted exec may still be queued there,
    // and HIP does not promise the lifetime of an in-flight exec, so the stream is drained first
    void insert(const std::vector<uint64_t>& key, std::vector<hipGraph_t> g, std::vector<hipGraphExec_t> x, hipStream_t s) {
        if (entries.size() >= cap) {
            size_t old = 0;
            for (size_t i = 1; i < entries.size(); ++i) if (entries[i].stamp < entries[old].stamp) old = i;
            (void)hipStreamSynchronize(s);
            destroy_(entries[old]);
            entries.erase(entries.begin() + old);
        }
        entries.push_back(Entry{key, std::move(g), std::move(x), ++clock});
    }
    void clear() {
        for (auto& e : entries) destroy_(e);
        entries.clear();
    }
};

// Run `enqueue()` (which only launches on `s`, and may call graph_cut(s)) through the cache: replay when `key` is known,
// otherwise capture + instantiate + launch.  `enqueue` returns 0 on success.
template <typename F>
int run_cached(GraphCache& cache, hipStream_t s, const std::vector<uint64_t>& key, F&& enqueue) {
    if (const std::vector<hipGraphExec_t>* xs = cache.find(key)) {
        for (hipGraphExec_t x : *xs) SSLAM_HIP_CHECK(hipGraphLaunch(x, s));
        return 0;
    }
    GraphCapture cap;
    cap.s = s;
    SSLAM_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    t_capture = &cap;
    const int rc = enqueue();
    t_capture = nullptr;
    bool ok = !cap.failed && capture_end_segment_(cap);
    if (cap.failed && !ok) {                                   // (a failed cut may have left the stream capturing: close it)
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
            hipGraph_t g = nullptr;
            (void)hipStreamEndCapture(s, &g);
            if (g) (void)hipGraphDestroy(g);
        }
    }
    if (rc || !ok) {
        for (auto x : cap.execs) (void)hipGraphExecDestroy(x);
        for (auto g : cap.graphs) (void)hipGraphDestroy(g);
        return rc ? rc : 1;
    }
    const std::vector<hipGraphExec_t> xs = cap.execs;
    cache.insert(key, std::move(cap.graphs), std::move(cap.execs), s);
    for (hipGraphExec_t x : xs) SSLAM_HIP_CHECK(hipGraphLaunch(x, s));
    return 0;
}

}  // namespace sslam

struct sslam_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // scratch for the *_host BA entry point (grown on demand, outside any graph)
    void* ba_scratch = nullptr;
    size_t ba_scratch_bytes = 0;
    // instances (ALIKED / LightGlue) created on this context; sslam_ctx_destroy with instances alive only marks
    // the context closed and the last instance's destroy frees it (a garbage collector may finalise a context
    // before its instances; their destroy still needs the stream)
    int instances = 0;
    bool closed = false;
};
namespace sslam {
void ctx_retain(sslam_ctx* ctx);
void ctx_release(sslam_ctx* ctx);      // frees a closed context when its last instance goes
}
