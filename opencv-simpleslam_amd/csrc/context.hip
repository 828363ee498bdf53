// context.hip - library/context entry points of the C-ABI (include/sslam_hip.h).
#include "common.hpp"

namespace sslam {
static thread_local std::string g_last_error;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
}  // namespace sslam

extern "C" {

int sslam_abi_version(void) { return 1; }

const char* sslam_last_error(void) { return sslam::g_last_error.c_str(); }

int sslam_device_count(int* n_out) {
    SSLAM_REQUIRE(n_out != nullptr, "sslam_device_count: n_out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *n_out = 0;
        sslam::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return 1;
    }
    *n_out = n;
    return 0;
}

int sslam_ctx_create(int device, void* stream, sslam_ctx** out) {
    SSLAM_REQUIRE(out != nullptr, "sslam_ctx_create: out is NULL");
    int n = 0;
    SSLAM_HIP_CHECK(hipGetDeviceCount(&n));
    SSLAM_REQUIRE(device >= 0 && device < n, "sslam_ctx_create: device %d out of range (%d visible)",
                  device, n);
    SSLAM_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    SSLAM_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    SSLAM_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
                  "sslam_ctx_create: device %d is %s; this library is built for gfx950 only",
                  device, prop.gcnArchName);
    sslam_ctx* c = new sslam_ctx();
    c->device = device;
    if (stream) {
        c->stream = (hipStream_t)stream;
        c->owns_stream = false;
    } else {
        SSLAM_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->owns_stream = true;
    }
    SSLAM_HIP_CHECK(hipEventCreate(&c->ev0));
    SSLAM_HIP_CHECK(hipEventCreate(&c->ev1));
    *out = c;
    return 0;
}

static void ctx_free(sslam_ctx* ctx);
extern "C++" {
namespace sslam {
void ctx_retain(sslam_ctx* ctx) { ++ctx->instances; }
void ctx_release(sslam_ctx* ctx) {
    if (--ctx->instances == 0 && ctx->closed) ctx_free(ctx);
}
}  // namespace sslam
}

int sslam_ctx_destroy(sslam_ctx* ctx) {
    if (!ctx) return 0;
    if (ctx->instances > 0) {           // instances alive: the last one's destroy frees the context
        ctx->closed = true;
        (void)hipStreamSynchronize(ctx->stream);
        return 0;
    }
    ctx_free(ctx);
    return 0;
}

static void ctx_free(sslam_ctx* ctx) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ba_scratch) (void)hipFree(ctx->ba_scratch);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int sslam_ctx_sync(sslam_ctx* ctx) {
    SSLAM_REQUIRE(ctx != nullptr, "sslam_ctx_sync: ctx is NULL");
    SSLAM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

void* sslam_ctx_stream(sslam_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int sslam_timer_start(sslam_ctx* ctx) {
    SSLAM_REQUIRE(ctx != nullptr, "sslam_timer_start: ctx is NULL");
    SSLAM_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
    return 0;
}

int sslam_timer_stop(sslam_ctx* ctx, float* elapsed_ms_out) {
    SSLAM_REQUIRE(ctx != nullptr && elapsed_ms_out != nullptr, "sslam_timer_stop: NULL argument");
    SSLAM_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
    SSLAM_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    SSLAM_HIP_CHECK(hipEventElapsedTime(elapsed_ms_out, ctx->ev0, ctx->ev1));
    return 0;
}

int sslam_malloc(sslam_ctx* ctx, size_t bytes, void** dptr_out) {
    SSLAM_REQUIRE(ctx != nullptr && dptr_out != nullptr, "sslam_malloc: NULL argument");
    SSLAM_HIP_CHECK(hipSetDevice(ctx->device));
    SSLAM_HIP_CHECK(hipMalloc(dptr_out, bytes ? bytes : 1));
    return 0;
}

int sslam_free(sslam_ctx* ctx, void* dptr) {
    SSLAM_REQUIRE(ctx != nullptr, "sslam_free: ctx is NULL");
    if (dptr) SSLAM_HIP_CHECK(hipFree(dptr));
    return 0;
}

int sslam_memcpy_h2d(sslam_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    SSLAM_REQUIRE(ctx != nullptr, "sslam_memcpy_h2d: ctx is NULL");
    SSLAM_HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    SSLAM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int sslam_memcpy_d2h(sslam_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    SSLAM_REQUIRE(ctx != nullptr, "sslam_memcpy_d2h: ctx is NULL");
    SSLAM_HIP_CHECK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SSLAM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

/* ---- stream ordering without any other GPU runtime (the frame pipeline chains extractor and
 * matcher contexts with these; nothing here synchronises the host) */
int sslam_event_create(sslam_ctx* ctx, void** event_out) {
    SSLAM_REQUIRE(ctx != nullptr && event_out != nullptr, "sslam_event_create: NULL argument");
    SSLAM_HIP_CHECK(hipSetDevice(ctx->device));
    hipEvent_t e;
    SSLAM_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event_out = (void*)e;
    return 0;
}

/* A TIMING event (hipEventDefault) and the elapsed time between two of them: the bench stamps the end of every
 * round of the running pipeline on a collector stream and reads the intervals afterwards - per-round durations
 * without a host synchronisation inside the timed region. */
int sslam_timing_event_create(sslam_ctx* ctx, void** event_out) {
    SSLAM_REQUIRE(ctx != nullptr && event_out != nullptr, "sslam_timing_event_create: NULL argument");
    SSLAM_HIP_CHECK(hipSetDevice(ctx->device));
    hipEvent_t e;
    SSLAM_HIP_CHECK(hipEventCreate(&e));
    *event_out = (void*)e;
    return 0;
}

int sslam_event_elapsed_ms(void* start_event, void* stop_event, float* ms_out) {
    SSLAM_REQUIRE(start_event && stop_event && ms_out, "sslam_event_elapsed_ms: NULL argument");
    SSLAM_HIP_CHECK(hipEventSynchronize((hipEvent_t)stop_event));
    SSLAM_HIP_CHECK(hipEventElapsedTime(ms_out, (hipEvent_t)start_event, (hipEvent_t)stop_event));
    return 0;
}

int sslam_event_destroy(void* event) {
    if (event) SSLAM_HIP_CHECK(hipEventDestroy((hipEvent_t)event));
    return 0;
}

int sslam_event_record(sslam_ctx* ctx, void* event) {
    SSLAM_REQUIRE(ctx != nullptr && event != nullptr, "sslam_event_record: NULL argument");
    SSLAM_HIP_CHECK(hipEventRecord((hipEvent_t)event, ctx->stream));
    return 0;
}

int sslam_ctx_wait_event(sslam_ctx* ctx, void* event) {
    SSLAM_REQUIRE(ctx != nullptr && event != nullptr, "sslam_ctx_wait_event: NULL argument");
    SSLAM_HIP_CHECK(hipStreamWaitEvent(ctx->stream, (hipEvent_t)event, 0));
    return 0;
}

int sslam_memcpy_d2d_async(sslam_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes) {
    SSLAM_REQUIRE(ctx != nullptr && (bytes == 0 || (dst_dev && src_dev)), "sslam_memcpy_d2d_async: NULL argument");
    if (bytes) SSLAM_HIP_CHECK(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

/* ---- page-locked host memory + enqueue-only host <-> device copies: the drop-in path stages the image and reads a
 * frame's {count, keypoints, descriptors} record back through these (a copy from / to pageable memory costs 30 - 80 us
 * of runtime staging per call and synchronises; measured in scripts/time_dropin_parts.py) */
int sslam_host_alloc(sslam_ctx* ctx, size_t bytes, void** hptr_out) {
    SSLAM_REQUIRE(ctx != nullptr && hptr_out != nullptr, "sslam_host_alloc: NULL argument");
    *hptr_out = nullptr;
    SSLAM_HIP_CHECK(hipSetDevice(ctx->device));
    SSLAM_HIP_CHECK(hipHostMalloc(hptr_out, bytes ? bytes : 1, hipHostMallocDefault));
    return 0;
}

int sslam_host_free(sslam_ctx* ctx, void* hptr) {
    SSLAM_REQUIRE(ctx != nullptr, "sslam_host_free: ctx is NULL");
    if (hptr) SSLAM_HIP_CHECK(hipHostFree(hptr));
    return 0;
}

int sslam_memcpy_h2d_async(sslam_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    SSLAM_REQUIRE(ctx != nullptr && (bytes == 0 || (dst_dev && src_host)), "sslam_memcpy_h2d_async: NULL argument");
    if (bytes) SSLAM_HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

int sslam_memcpy_d2h_async(sslam_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    SSLAM_REQUIRE(ctx != nullptr && (bytes == 0 || (dst_host && src_dev)), "sslam_memcpy_d2h_async: NULL argument");
    if (bytes) SSLAM_HIP_CHECK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}

int sslam_memset_async(sslam_ctx* ctx, void* dst_dev, int value, size_t bytes) {
    SSLAM_REQUIRE(ctx != nullptr && (bytes == 0 || dst_dev), "sslam_memset_async: NULL argument");
    if (bytes) SSLAM_HIP_CHECK(hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
    return 0;
}

}  // extern "C"
