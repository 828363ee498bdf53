// Does VALU work issue under a running MFMA, and does it depend on which register file the MFMA's
// operands live in?  One wave per SIMD (512-register budget), a loop of { 1 MFMA ; F independent v_fma },
// four independent accumulators in rotation.  Variants: C/D in arch VGPRs vs AGPRs, A/B in VGPRs vs AGPRs.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define MFV(c_) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c_) : "v"(a), "v"(b))

template <int CD_AGPR, int AB_AGPR, int F>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, long long* cyc) {
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = r; c1[r] = r + 1; c2[r] = r + 2; c3[r] = r + 3; }
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    const float m = 1.0001f, ad = 0.5f;
    const long long t0 = __builtin_amdgcn_s_memtime();
#define MF(c_) do { \
        if (CD_AGPR && AB_AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c_) : "a"(a), "a"(b)); \
        else if (CD_AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c_) : "v"(a), "v"(b)); \
        else if (AB_AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c_) : "a"(a), "a"(b)); \
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c_) : "v"(a), "v"(b)); } while (0)
#define FILL() do { _Pragma("unroll") for (int f = 0; f < F; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[f & 7]) : "v"(m), "v"(ad)); } while (0)
    for (int it = 0; it < iters; ++it) {
        MF(c0); FILL(); MF(c1); FILL(); MF(c2); FILL(); MF(c3); FILL();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// two waves per SIMD (512-thread workgroup, 256-register budget): the same loop in both waves
template <int F>
__global__ __launch_bounds__(512, 2) void k2(float* out, int iters, long long* cyc) {
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = r; c1[r] = r + 1; c2[r] = r + 2; c3[r] = r + 3; }
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    const float m = 1.0001f, ad = 0.5f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        MFV(c0); FILL(); MFV(c1); FILL(); MFV(c2); FILL(); MFV(c3); FILL();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int F>
void run2(float* out, long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k2<F>), dim3(256), dim3(512), 0, 0, out, iters, cyc);
    hipLaunchKernelGGL((k2<F>), dim3(256), dim3(512), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    long long hh[8]; hipMemcpy(hh, cyc, 64, hipMemcpyDeviceToHost);
    long long h = 0, lo = 1LL << 62; for (int i = 0; i < 8; ++i) { h = hh[i] > h ? hh[i] : h; lo = hh[i] < lo ? hh[i] : lo; }
    printf("TWO waves per SIMD, %d fillers per MFMA: slowest wave %.1f cycles per own MFMA = %.1f per MFMA on the pipe (fastest wave %.1f)\n", F,
           (double)h / (iters * 4.0), (double)h / (iters * 8.0), (double)lo / (iters * 4.0));
}

// one wave per SIMD, filler KIND: 0 independent v_fma, 1 ONE dependent v_fma chain, 2 v_exp_f32 (independent),
// 3 two dependent chains, 4 v_max3_f32 (independent), 5 v_cvt_pk_f16_f32 (independent)
template <int KIND, int F>
__global__ __launch_bounds__(256, 1) void k3(float* out, int iters, long long* cyc) {
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = r; c1[r] = r + 1; c2[r] = r + 2; c3[r] = r + 3; }
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    const float m = 0.9999f, ad = 0.5f;
    const long long t0 = __builtin_amdgcn_s_memtime();
#define FILLK() do { _Pragma("unroll") for (int f = 0; f < F; ++f) { \
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[f & 7]) : "v"(m), "v"(ad)); \
        else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(m), "v"(ad)); \
        else if (KIND == 2) asm volatile("v_exp_f32 %0, %1" : "=v"(v[f & 7]) : "v"(v[(f + 4) & 7])); \
        else if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[f & 1]) : "v"(m), "v"(ad)); \
        else if (KIND == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[f & 7]) : "v"(m), "v"(ad)); \
        else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v[f & 7]) : "v"(m), "v"(ad)); } } while (0)
    for (int it = 0; it < iters; ++it) {
        MFV(c0); FILLK(); MFV(c1); FILLK(); MFV(c2); FILLK(); MFV(c3); FILLK();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND, int F>
void run3(float* out, long long* cyc, const char* what) {
    const int iters = 2000;
    hipLaunchKernelGGL((k3<KIND, F>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipLaunchKernelGGL((k3<KIND, F>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("one wave, %d x %-34s per MFMA: %.1f cycles per MFMA\n", F, what, (double)h / (iters * 4.0));
}

template <int CD, int AB, int F>
void run(float* out, long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<CD, AB, F>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipLaunchKernelGGL((k<CD, AB, F>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("C/D in %s, A/B in %s, %d fillers per MFMA: %.1f cycles per MFMA (s_memtime ticks)\n", CD ? "AGPR" : "VGPR", AB ? "AGPR" : "VGPR", F,
           (double)h / (iters * 4.0));
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64);
    run<0, 0, 0>(out, cyc); run<0, 0, 3>(out, cyc); run<0, 0, 5>(out, cyc); run<0, 0, 8>(out, cyc);
    run<1, 0, 0>(out, cyc); run<1, 0, 3>(out, cyc); run<1, 0, 5>(out, cyc); run<1, 0, 8>(out, cyc);
    run<1, 1, 0>(out, cyc); run<1, 1, 3>(out, cyc); run<1, 1, 5>(out, cyc); run<1, 1, 8>(out, cyc);
    run<0, 1, 5>(out, cyc);
    run3<1, 3>(out, cyc, "v_fma in ONE dependent chain"); run3<1, 5>(out, cyc, "v_fma in ONE dependent chain");
    run3<3, 4>(out, cyc, "v_fma in TWO dependent chains"); run3<3, 6>(out, cyc, "v_fma in TWO dependent chains");
    run3<2, 1>(out, cyc, "v_exp_f32"); run3<2, 2>(out, cyc, "v_exp_f32"); run3<2, 4>(out, cyc, "v_exp_f32");
    run3<4, 5>(out, cyc, "v_max3_f32"); run3<5, 5>(out, cyc, "v_cvt_pk_f16_f32");
    hipFree(out); hipMalloc(&out, 256 * 512 * 4);
    run2<0>(out, cyc); run2<3>(out, cyc); run2<5>(out, cyc); run2<6>(out, cyc); run2<7>(out, cyc); run2<8>(out, cyc); run2<10>(out, cyc); run2<12>(out, cyc);
    return 0;
}
