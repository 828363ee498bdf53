// agg_victim.hip - al_aggregate_kernel ALONE, in a loop on fixed inputs, every launch compared with the first one on the device.
//
// The experiments of profiles/r06_aggregate_rnorm_diagnosis.md ran the whole extractor (32 launches per call) to see the one
// fault of this repository (1 / ||F|| missing one gather's term in lanes 48..63, only in the packed-fp32 code shape, only with
// other queues' kernels on the GPU).  This file asks the next question: is the kernel BY ITSELF, with nothing of the extractor
// around it, enough of a victim?  It includes the product's source as it lies (no copy), so the kernel is the very code the
// library ships, compiled with whatever -DAL_AGG_* the build line gives (scripts/agg_victim.sh):
//     -DAL_AGG_FAST_SELU=2 -DAL_AGG_PACKED=1      the failing shape (147 v_pk_*)
//     (nothing)                                     the product's shape (packed fp32 off)
// C entry points (ctypes, scripts/agg_victim_run.py): victim_create / victim_run / victim_poll / victim_destroy.
#include "../../opencv-simpleslam_amd/csrc/aliked_kernels.hip"

// (the three functions of the library's other translation units that the included source refers to; nothing here calls them)
namespace sslam {
void set_error(const char*, ...) {}
void ctx_retain(sslam_ctx*) {}
void ctx_release(sslam_ctx*) {}
}

namespace {

struct Victim {
    Pyr P; float *ws0, *w1, *s8, *rnorm, *ref_rnorm, *ref_s8; unsigned* bad;      // bad[0] rnorm mismatches, bad[1] s8 mismatches, bad[2..] first events
    int Hp, Wp, F; size_t fs; bool have_ref; std::vector<void*> owned;
};

// bad[0] += rnorm words that differ, bad[1] += s8 words that differ; the first 15 differing rnorm words in full:
// bad[4 + 4k] = index, +1 = got, +2 = reference, +3 = launch number
__global__ void victim_compare(const float* __restrict__ rn, const float* __restrict__ ref, size_t n, const float* __restrict__ s8,
                               const float* __restrict__ ref8, size_t n8, unsigned* __restrict__ bad, unsigned launch) {
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = i0; i < n; i += step) {
        const unsigned a = __float_as_uint(rn[i]), b = __float_as_uint(ref[i]);
        if (a != b) {
            const unsigned k = atomicAdd(&bad[0], 1u);
            if (k < 15) { bad[4 + 4 * k] = (unsigned)i; bad[5 + 4 * k] = a; bad[6 + 4 * k] = b; bad[7 + 4 * k] = launch; }
        }
    }
    for (size_t i = i0; i < n8; i += step)
        if (__float_as_uint(s8[i]) != __float_as_uint(ref8[i])) atomicAdd(&bad[1], 1u);
}

// a patched build of the kernel (scripts/agg_isa_patch.py: the compiler's assembly with one property changed by hand), loaded as a
// code object and launched instead of the compiled-in kernel when victim_use_module() has been called
hipFunction_t g_patched = nullptr;

float* dev_fill(Victim* v, size_t n, unsigned seed, float lo, float hi) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = lo + (hi - lo) * (float)(s >> 8) * (1.0f / 16777216.0f); }
    float* d = nullptr;
    if (hipMalloc(&d, n * sizeof(float)) != hipSuccess) return nullptr;
    hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    v->owned.push_back(d);
    return d;
}
float* dev_zero(Victim* v, size_t n) {
    float* d = nullptr;
    if (hipMalloc(&d, n * sizeof(float)) != hipSuccess) return nullptr;
    hipMemset(d, 0, n * sizeof(float));
    v->owned.push_back(d);
    return d;
}

}  // namespace

extern "C" {

// Hp, Wp multiples of 32 (the padded network size; 384 x 1248 for a KITTI frame), F frames per launch
void* victim_create(int Hp, int Wp, int F, unsigned seed) {
    if (Hp % 32 || Wp % 32 || F < 1 || F > 8) return nullptr;
    Victim* v = new Victim();
    v->Hp = Hp; v->Wp = Wp; v->F = F; v->have_ref = false;
    const size_t HW = (size_t)Hp * Wp;
    v->fs = 32 * HW;                                       // one frame stride for every buffer, as in the product
    const size_t tot = v->fs * F;
    float* x1 = dev_fill(v, tot, seed + 1, -1.0f, 1.0f);
    v->w1 = dev_fill(v, 16 * 32, seed + 2, -0.4f, 0.4f);
    v->ws0 = dev_fill(v, 128 * 8, seed + 3, -0.3f, 0.3f);
    float* pre2 = dev_fill(v, tot, seed + 4, 0.1f, 1.0f);   // Gram maps: positive, every term of the quadratic form visible
    float* pre3 = dev_fill(v, tot, seed + 5, 0.1f, 1.0f);
    float* pre4 = dev_fill(v, tot, seed + 6, 0.1f, 1.0f);
    float* g1cl = dev_zero(v, tot);
    v->s8 = dev_zero(v, tot); v->rnorm = dev_zero(v, tot); v->ref_rnorm = dev_zero(v, tot); v->ref_s8 = dev_zero(v, tot);
    v->bad = (unsigned*)dev_zero(v, 64);
    if (!x1 || !v->w1 || !v->ws0 || !pre2 || !pre3 || !pre4 || !g1cl || !v->s8 || !v->rnorm || !v->ref_rnorm || !v->ref_s8 || !v->bad) return nullptr;
    Pyr P{x1, nullptr, nullptr, nullptr, v->w1, Hp, Wp, g1cl};
    auto step = [](int full, int S) { return (float)(full / S - 1) / (float)(full - 1); };
    P.sy2 = step(Hp, 2); P.sx2 = step(Wp, 2); P.sy8 = step(Hp, 8); P.sx8 = step(Wp, 8); P.sy32 = step(Hp, 32); P.sx32 = step(Wp, 32);
    P.g2cl = P.g3cl = P.g4cl = nullptr;
    P.pre2 = pre2; P.pre3 = pre3; P.pre4 = pre4;
    v->P = P;
    hipDeviceSynchronize();
    return v;
}

// `iters` launches of the kernel on `stream`, each followed by the comparison with the first launch's output (which the first call
// of this function produces and keeps).  Nothing waits on the host.  check_every: compare after every n-th launch only.
int victim_run(void* h, void* stream, int iters, int check_every) {
    Victim* v = (Victim*)h;
    hipStream_t s = (hipStream_t)stream;
    const size_t HW = (size_t)v->Hp * v->Wp;
    static unsigned launch = 0;
    for (int i = 0; i < iters; ++i) {
        if (g_patched) {
            const float* ws0 = v->ws0;
            void* args[] = {&v->P, &ws0, &v->s8, &v->rnorm, &v->fs};
            if (hipModuleLaunchKernel(g_patched, sslam::cdiv(v->Wp, 256), v->Hp, v->F, 256, 1, 1, 0, s, args, nullptr) != hipSuccess) return -2;
        } else {
            hipLaunchKernelGGL(al_aggregate_kernel, dim3(sslam::cdiv(v->Wp, 256), v->Hp, v->F), dim3(256), 0, s, v->P, v->ws0, v->s8, v->rnorm, v->fs);
        }
        ++launch;
        if (!v->have_ref) {
            hipMemcpyAsync(v->ref_rnorm, v->rnorm, v->fs * v->F * sizeof(float), hipMemcpyDeviceToDevice, s);
            hipMemcpyAsync(v->ref_s8, v->s8, v->fs * v->F * sizeof(float), hipMemcpyDeviceToDevice, s);
            v->have_ref = true;
        } else if (check_every > 0 && (i % check_every) == check_every - 1) {
            // per frame: rnorm [HW] at f * fs, s8 [8][HW] at f * fs
            for (int f = 0; f < v->F; ++f)
                hipLaunchKernelGGL(victim_compare, dim3(512), dim3(256), 0, s, v->rnorm + f * v->fs, v->ref_rnorm + f * v->fs, HW,
                                   v->s8 + f * v->fs, v->ref_s8 + f * v->fs, 8 * HW, v->bad, launch);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// from here on victim_run launches `kernel_name` of the code object at `path` (same signature as al_aggregate_kernel)
int victim_use_module(const char* path, const char* kernel_name) {
    hipModule_t m = nullptr;
    if (hipModuleLoad(&m, path) != hipSuccess) return -1;
    if (hipModuleGetFunction(&g_patched, m, kernel_name) != hipSuccess) { g_patched = nullptr; return -2; }
    return 0;
}

// waits for the stream and copies the 64 counter words out
int victim_poll(void* h, void* stream, unsigned* out64) {
    Victim* v = (Victim*)h;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    return hipMemcpy(out64, v->bad, 64 * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

void victim_destroy(void* h) {
    Victim* v = (Victim*)h;
    for (void* p : v->owned) hipFree(p);
    delete v;
}

}  // extern "C"
