// aggressors.hip - r06, profiles/r06_aggregate_rnorm_diagnosis.md: synthetic kernels to run BESIDE the unstable code shape of
// al_aggregate_kernel (scripts/diag_agg_rnorm.py <repeats> 1 <kind>), one property each, to find which one the events need.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/ubench/libaggr.so scripts/ubench/aggressors.hip
#include <hip/hip_runtime.h>

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_trans(float* out, int iters) {                 // v_exp_f32 only
    float x = threadIdx.x * 1e-3f, acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        float e;
        asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(x));
        acc += e; x = x * 0.999f + 1e-4f;
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {                  // matrix pipe only
    half4_t a = {1, 2, 3, 4}, b = {4, 3, 2, 1};
    f32x16_t c = {};
    for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c, 0, 0, 0);
    if (c[0] == 123.456f) out[0] = c[3];
}
__global__ __launch_bounds__(256) void k_pk(float* out, int iters) {                    // packed fp32 VALU only
    f32x2_t a = {1.0001f, 0.9999f}, b = {threadIdx.x * 1e-6f, 1e-7f}, c = {0, 0};
    for (int i = 0; i < iters; ++i) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(c) : "v"(a), "v"(b));
    if (c[0] == 123.456f) out[0] = c[1];
}
__global__ __launch_bounds__(256) void k_valu(float* out, int iters) {                  // plain fp32 VALU only
    float a = 1.0001f, b = threadIdx.x * 1e-6f, c = 0;
    for (int i = 0; i < iters; ++i) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(c) : "v"(a), "v"(b));
    if (c == 123.456f) out[0] = c;
}
__global__ __launch_bounds__(256) void k_lds(float* out, int iters) {                   // LDS traffic (48 KB allocated)
    __shared__ float buf[12288];
    for (int i = threadIdx.x; i < 12288; i += 256) buf[i] = i;
    __syncthreads();
    float acc = 0; unsigned j = threadIdx.x;
    for (int i = 0; i < iters; ++i) { acc += buf[j % 12288]; buf[(j * 7 + 3) % 12288] = acc; j = j * 5 + 1; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ tab, float* out, int iters, unsigned mask) {   // L1 / L2-resident vector loads
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    float acc = 0;
    for (int i = 0; i < iters; ++i) { h = h * 1664525u + 1013904223u; acc += tab[(h >> 8) & mask]; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_store(float* dst, int iters, unsigned mask) {   // vector stores into an L2-resident window
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    for (int i = 0; i < iters; ++i) { h = h * 1664525u + 1013904223u; dst[(h >> 8) & mask] = (float)i; }
}
__global__ __launch_bounds__(256) void k_scalar(const int* __restrict__ tab, float* out, int iters) {                      // scalar loads (wave-uniform)
    int acc = 0;
    for (int i = 0; i < iters; ++i) acc += tab[(blockIdx.x * 131 + i * 17) & 4095];
    if (acc == 123456789) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_ldsdma(const float* __restrict__ tab, float* out, int iters, unsigned mask) {   // LDS-DMA: global_load_lds_dwordx4
    extern __shared__ float dyn[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned h = (blockIdx.x * 4 + wave) * 2654435761u;
    float acc = 0;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const float* src = tab + ((((h >> 8) & mask) & ~255u) + lane * 4);            // the wave's 1 KiB piece, 16 B per lane
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dyn + wave * 2048 + (i & 7) * 256), 16, 0, 0);
        if ((i & 7) == 7) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += dyn[wave * 2048 + lane];
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

static float* g_buf = nullptr;
extern "C" int aggr_launch(int kind, void* stream, int blocks, int iters) {
    hipStream_t s = (hipStream_t)stream;
    if (!g_buf) { if (hipMalloc(&g_buf, 64 << 20) != hipSuccess) return 1; (void)hipMemset(g_buf, 0, 64 << 20); }
    float* out = g_buf + (60 << 18);
    switch (kind) {
        case 1: hipLaunchKernelGGL(k_trans, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 2: hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 3: hipLaunchKernelGGL(k_pk, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 4: hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 5: hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 6: hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, s, g_buf, out, iters, (1u << 20) - 1); break;   // 4 MB window
        case 7: hipLaunchKernelGGL(k_store, dim3(blocks), dim3(256), 0, s, g_buf, iters, (1u << 20) - 1); break;
        case 8: hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, s, (const int*)g_buf, out, iters); break;
        case 9: hipLaunchKernelGGL(k_ldsdma, dim3(blocks), dim3(256), 32768, s, g_buf, out, iters, (1u << 20) - 1); break;
        default: return 2;
    }
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
