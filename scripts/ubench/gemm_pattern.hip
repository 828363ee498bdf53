// micro-benchmark: the ring GEMM's exact load pattern (64x128 tile, K = 512, k-panel planes), no math
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int M = 4096, N = 512, K = 512, BM = 64, BN = 128, KT = K / 64;
template <int NSTAGE, bool SWZ, bool COMPUTE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ A, const char* __restrict__ W, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int rb, cb;
    const int gx = N / BN;
    if (SWZ) { const int b = blockIdx.y * gridDim.x + blockIdx.x; const int xcd = b & 7, idx = b >> 3; rb = xcd + 8 * (idx / gx); cb = idx % gx; }
    else { rb = blockIdx.y; cb = blockIdx.x; }
    constexpr int STAGE = (2 * BM + 2 * BN) * 128, NI = (2 * BM + 2 * BN) / 8 / 4;   // 12
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int g = wave + 4 * j;                 // 0..47: [A hi 8][A lo 8][W hi 16][W lo 16]
            const char* s;
            if (g < 16) { const int plane = g / 8, grp = g % 8;
                s = A + (size_t)plane * M * K * 2 + ((size_t)kt * M + rb * BM + grp * 8) * 128 + lane * 16; }
            else { const int plane = (g - 16) / 16, grp = (g - 16) % 16;
                s = W + (size_t)plane * N * K * 2 + ((size_t)kt * N + cb * BN + grp * 8) * 128 + lane * 16; }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                             (__attribute__((address_space(3))) void*)(lds + stage * STAGE + g * 1024), 16, 0, 0);
        }
    };
    float acc = 0.f;
    for (int p = 0; p < NSTAGE - 1; ++p) issue(p, p);
    for (int kt = 0; kt < KT; ++kt) {
        const int younger = min(KT - 1 - kt, NSTAGE - 2);
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + NSTAGE - 1 < KT) issue(kt + NSTAGE - 1, (kt + NSTAGE - 1) % NSTAGE);
        if (COMPUTE) { for (int i = 0; i < 48; ++i) acc += *reinterpret_cast<float*>(lds + (kt % NSTAGE) * STAGE + ((t * 16 + i * 4096) % STAGE)); }
        else acc += *reinterpret_cast<float*>(lds + (kt % NSTAGE) * STAGE + t * 16);
    }
    if (acc == 12345.f) out[0] = acc;
}
template <int NS, bool SWZ, bool C>
int run(const char* name, const char* A, const char* W, float* out) {
    size_t lds = (size_t)NS * (2 * BM + 2 * BN) * 128;
    CK(hipFuncSetAttribute((const void*)k<NS, SWZ, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    dim3 grid(N / BN, M / BM);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NS, SWZ, C>), grid, dim3(256), lds, 0, A, W, out);
    CK(hipEventRecord(a));
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k<NS, SWZ, C>), grid, dim3(256), lds, 0, A, W, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-34s %6.1f us/launch (%.0f MB requested -> %.2f TB/s)\n", name, ms * 50, 256.0 * KT * 48 / 1024, 256.0 * KT * 48e3 / (ms * 50) * 1e-6 );
    return 0;
}
int main() {
    char *A, *W; float* out;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&W, (size_t)N * K * 4)); CK(hipMalloc(&out, 64));
    CK(hipMemset(A, 1, (size_t)M * K * 4)); CK(hipMemset(W, 1, (size_t)N * K * 4));
    run<2, false, false>("ring2 no-swizzle", A, W, out);
    run<2, true, false>("ring2 xcd-swizzle", A, W, out);
    run<3, true, false>("ring3 xcd-swizzle", A, W, out);
    run<2, true, true>("ring2 xcd-swizzle + lds reads", A, W, out);
    return 0;
}
