// vmem_addr_war.hip - probe (r05, profiles/r05_aggregate_selu_hazard.md): can a VALU instruction that overwrites the address
// VGPR pair of a global_load ONE instruction after the load was issued change what the load fetches?
//
// The failing builds of al_aggregate_kernel are full of the sequence
//     global_load_dword vA, v[R:R+1], off
//     v_lshl_add_u64    v[R:R+1], s[..], 0, v[..]      ; the next address, into the same pair
//     global_load_dword vB, v[R:R+1], off
// (57 places against 4 in the build that never fails) and their faults are one 16-lane group of ONE gathered value, only with
// other kernels on the GPU.  This program issues exactly that sequence in inline assembly in a loop, on arrays whose elements
// hold their own index, and counts loads that came back with another element - alone, and beside memory-bound and
// transcendental-bound kernels on other streams.  MODE 0: the sequence as above; 1: with `s_nop 0` between load and overwrite.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/vmem_addr_war scripts/ubench/vmem_addr_war.hip && /tmp/vmem_addr_war [seconds=20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ A, const float* __restrict__ B, unsigned n_mask, int iters,
                                             unsigned long long* bad, unsigned long long* done) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned h = tid * 2654435761u + 12345u;
    unsigned long long nbad = 0;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        // a wave's 64 lanes read 64 consecutive floats (one 256-byte run = four 64-byte beats) at a pseudo-random place
        const unsigned ia = (((h >> 8) & n_mask) & ~63u) + (threadIdx.x & 63);
        const unsigned ib = ((((h >> 8) * 7u + 64u * 977u) & n_mask) & ~63u) + (threadIdx.x & 63);
        unsigned long long addr = (unsigned long long)(A + ia);
        const unsigned long long offb = (unsigned long long)ib * 4ull;
        float va, vb;
        if (MODE == 0)
            asm volatile("global_load_dword %0, %2, off\n\t"
                         "v_lshl_add_u64 %2, %3, 0, %4\n\t"
                         "global_load_dword %1, %2, off\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(va), "=&v"(vb), "+v"(addr) : "s"(B), "v"(offb) : "memory");
        else
            asm volatile("global_load_dword %0, %2, off\n\t"
                         "s_nop 0\n\t"
                         "v_lshl_add_u64 %2, %3, 0, %4\n\t"
                         "global_load_dword %1, %2, off\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(va), "=&v"(vb), "+v"(addr) : "s"(B), "v"(offb) : "memory");
        nbad += (va != (float)ia) + (vb != -(float)ib);
    }
    if (nbad) atomicAdd(bad, nbad);
    if (threadIdx.x == 0) atomicAdd(done, (unsigned long long)iters * 2ull * blockDim.x);
}

__global__ void hog_memory(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {          // HBM-bound copy
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void hog_trans(float* out, int iters) {                                                           // v_exp_f32-bound
    float x = threadIdx.x * 1e-3f, acc = 0.0f;
    for (int i = 0; i < iters; ++i) { acc += __expf(x); x = x * 0.999f + 1e-4f; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void fill(float* p, size_t n, float sign) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = sign * (float)i;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 20.0;
    const size_t N = 1u << 24;                       // 16 M floats per array: indices are exact in fp32 (< 2^24)
    float *A, *B, *scratch;
    float4 *h0, *h1;
    unsigned long long *cnt;
    const size_t HN = 1u << 26;                      // 1 GiB per hog buffer
    CHECK(hipMalloc(&A, N * 4)); CHECK(hipMalloc(&B, N * 4)); CHECK(hipMalloc(&scratch, 4096));
    CHECK(hipMalloc(&h0, HN * 16)); CHECK(hipMalloc(&h1, HN * 16)); CHECK(hipMalloc(&cnt, 64));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, A, N, 1.0f);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, B, N, -1.0f);
    CHECK(hipMemset(h0, 0, HN * 16));
    CHECK(hipDeviceSynchronize());
    hipStream_t sp, sm, st;
    CHECK(hipStreamCreate(&sp)); CHECK(hipStreamCreate(&sm)); CHECK(hipStreamCreate(&st));
    const char* names[] = {"alone", "beside an HBM-bound copy", "beside a v_exp_f32-bound kernel", "beside both"};
    for (int mode = 0; mode < 2; ++mode)
        for (int env = 0; env < 4; ++env) {
            CHECK(hipMemset(cnt, 0, 64));
            const auto t0 = std::chrono::steady_clock::now();
            int rounds = 0;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds / 8.0) {
                if (env & 1) hipLaunchKernelGGL(hog_memory, dim3(2048), dim3(256), 0, sm, h0, h1, HN);
                if (env & 2) hipLaunchKernelGGL(hog_trans, dim3(1024), dim3(256), 0, st, scratch, 200000);
                for (int k = 0; k < 8; ++k) {
                    if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(1024), dim3(256), 0, sp, A, B, (unsigned)(N - 1), 2000, cnt, cnt + 1);
                    else hipLaunchKernelGGL(probe<1>, dim3(1024), dim3(256), 0, sp, A, B, (unsigned)(N - 1), 2000, cnt, cnt + 1);
                }
                CHECK(hipStreamSynchronize(sp));
                ++rounds;
            }
            CHECK(hipDeviceSynchronize());
            unsigned long long h[2];
            CHECK(hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost));
            printf("%-34s %s: %llu wrong of %.3g loads (%d rounds)\n", mode ? "s_nop 0 between load and overwrite" : "overwrite right behind the load", names[env],
                   h[0], (double)h[1], rounds);
            fflush(stdout);
        }
    return 0;
}
