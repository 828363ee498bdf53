// micro-benchmark: per-CU streaming rate of global_load_lds_dwordx4 tiles (32 KB per tile, 4 waves)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NSTAGE, bool USE_DMA>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ src, size_t per_block_stride, int tiles,
                                                     float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const char* base = src + (size_t)blockIdx.x * per_block_stride;
    constexpr int TILE = 32768, NI = TILE / 1024 / 4;     // 8 DMA per wave per tile
    float acc = 0.f;
    auto issue = [&](int tile, int stage) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int g = wave + 4 * j;
            const char* s = base + (size_t)tile * TILE + g * 1024 + lane * 16;
            char* d = lds + stage * TILE + g * 1024;
            if constexpr (USE_DMA) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                                 (__attribute__((address_space(3))) void*)d, 16, 0, 0);
            } else {
                *reinterpret_cast<float4*>(d + lane * 16) = *reinterpret_cast<const float4*>(s);
            }
        }
    };
    for (int p = 0; p < NSTAGE - 1; ++p) if (p < tiles) issue(p, p);
    for (int kt = 0; kt < tiles; ++kt) {
        if constexpr (USE_DMA) {
            const int younger = min(tiles - 1 - kt, NSTAGE - 2);
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            __syncthreads();
        }
        if (kt + NSTAGE - 1 < tiles) issue(kt + NSTAGE - 1, (kt + NSTAGE - 1) % NSTAGE);
        acc += *reinterpret_cast<float*>(lds + (kt % NSTAGE) * TILE + t * 16);
    }
    if (acc == 12345.f) out[0] = acc;
}

template <int NSTAGE, bool DMA>
int run(const char* name, const char* d, size_t stride, int tiles, int blocks, float* out) {
    size_t lds = (size_t)NSTAGE * 32768;
    CK(hipFuncSetAttribute((const void*)stream_kernel<NSTAGE, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stream_kernel<NSTAGE, DMA>), dim3(blocks), dim3(256), lds, 0, d, stride, tiles, out);
    CK(hipEventRecord(a));
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((stream_kernel<NSTAGE, DMA>), dim3(blocks), dim3(256), lds, 0, d, stride, tiles, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double us = ms * 1e3 / reps, bytes = (double)blocks * tiles * 32768;
    printf("%-28s blocks %4d tiles %3d stride %8zu: %7.1f us/launch  %7.1f GB/s/CU  %6.2f TB/s total\n", name, blocks, tiles, stride,
           us, bytes / blocks / us * 1e-3, bytes / us * 1e-6);
    return 0;
}

int main() {
    char* d; float* out;
    const size_t total = (size_t)1 << 30;
    CK(hipMalloc(&d, total)); CK(hipMemset(d, 1, total)); CK(hipMalloc(&out, 64));
    for (int tiles : {8, 32}) {
        // same 256 KB for every block (L2-resident after first touch)
        run<2, true>("dma ring2 shared", d, 0, tiles, 256, out);
        run<4, true>("dma ring4 shared", d, 0, tiles, 256, out);
        // distinct region per block (8 MB footprint at 8 tiles -> L2/MALL), 64 MB at 32 tiles
        run<2, true>("dma ring2 distinct", d, (size_t)tiles * 32768, tiles, 256, out);
        run<4, true>("dma ring4 distinct", d, (size_t)tiles * 32768, tiles, 256, out);
        run<2, false>("regs ring2 distinct", d, (size_t)tiles * 32768, tiles, 256, out);
        run<4, true>("dma ring4 distinct 64blk", d, (size_t)tiles * 32768, tiles, 64, out);
    }
    return 0;
}
