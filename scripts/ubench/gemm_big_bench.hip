// micro-benchmark + correctness check of gemm_mainloop_big (csrc/gemm_f16x3_big.hpp) against the
// ring GEMM of gemm_f16x3.hpp on the batched LightGlue shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I opencv-simpleslam_amd/csrc scripts/ubench/gemm_big_bench.hip -o /tmp/gemm_big && /tmp/gemm_big
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <type_traits>
#include "gemm_f16x3_big_r02.hpp"   // r02 forms (128 x 256 / 8 waves, pipelined main loops, ablations); the product header keeps one
using namespace sslam;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int BM, int BN, int WM, int WN, int V, int NS>
__global__ __launch_bounds__(WM * WN * 64, 2) void k_big(SplitPtr A, SplitPtr W, float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    // XCD-aware order: all column tiles of a row block on one XCD, adjacent in dispatch order
    int rb, cb;
    {
        const int b = blockIdx.y * gridDim.x + blockIdx.x;
        if ((gridDim.y & 7) == 0) { const int xcd = b & 7, idx = b >> 3; rb = xcd + 8 * (idx / gridDim.x); cb = idx % gridDim.x; }
        else { rb = blockIdx.y; cb = blockIdx.x; }
    }
    GemmAH ga{A, A, K, K};
    f32x16 c1[TM][TN], c2[TM][TN];
    gemm_mainloop_big<BM, BN, WM, WN, V, NS>(ga, W, M, K, rb * BM, M, cb * BN, N, smem, c1, c2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave / WN, wn = wave % WN;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb * BM + wm * 32 * TM + i * 32 + acc_row(r, lane);
                const int col = cb * BN + wn * 32 * TN + j * 32 + (lane & 31);
#if GEMM_ABL & 8
                if (c1[i][j][r] == 123.456f)
#endif
                C[(size_t)row * N + col] = c1[i][j][r] + c2[i][j][r] * SPLIT_INV;
            }
}

// ---- experiment: 128 x 128 tile, 4 waves, FULL-LINE k-tiles (64 halves = 128 B per row and plane: one
// L2 request per line instead of two), 2 stages of 64 KB, one workgroup per CU
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void gemm_mainloop_wide(const GemmAH& ga, SplitPtr W, int a_rows, int K, int row0, int row_cap,
                                                   int col0, int col_cap, _Float16* smem,
                                                   f32x16 (&acc1)[BM / (32 * WM)][BN / (32 * WN)],
                                                   f32x16 (&acc2)[BM / (32 * WM)][BN / (32 * WN)]) {
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN), NWAVE = WM * WN, WBK = 64;
    constexpr int STAGE = (2 * BM + 2 * BN) * WBK;
    constexpr int PA = BM / 8, PW = BN / 8, JA = PA / NWAVE, JW = PW / NWAVE;
    static_assert(PA % NWAVE == 0 && PW % NWAVE == 0, "pieces divide");
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nkt = K / WBK;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc1[i][j][r] = 0.0f; acc2[i][j][r] = 0.0f; }
    const int prow = lane >> 3, pc = lane & 7;
    int aoff[JA], woff[JW];
#pragma unroll
    for (int q = 0; q < JA; ++q) {
        const int r = (wave + NWAVE * q) * 8 + prow;
        aoff[q] = min(row0 + r, row_cap - 1) * 64 + ((pc ^ ((r >> 1) & 7)) * 8);
    }
#pragma unroll
    for (int q = 0; q < JW; ++q) {
        const int r = (wave + NWAVE * q) * 8 + prow;
        woff[q] = min(col0 + r, col_cap - 1) * 64 + ((pc ^ ((r >> 1) & 7)) * 8);
    }
    auto issue = [&](int kt, int stage) {
        const int k = kt * WBK;
        const bool first = k < ga.K0;
        const int ka = first ? k : k - ga.K0;
        const size_t apan = (size_t)(ka >> 6) * a_rows * 64;
        const _Float16* pah = (first ? ga.A0.hi : ga.A1.hi) + apan;
        const _Float16* pal = (first ? ga.A0.lo : ga.A1.lo) + apan;
        const size_t wpan = (size_t)(k >> 6) * col_cap * 64;
        _Float16* sb = smem + (size_t)stage * STAGE + wave * 8 * WBK;
#pragma unroll
        for (int q = 0; q < JA; ++q) glds16_(pah + aoff[q], sb + q * NWAVE * 8 * WBK);
#pragma unroll
        for (int q = 0; q < JA; ++q) glds16_(pal + aoff[q], sb + BM * WBK + q * NWAVE * 8 * WBK);
#pragma unroll
        for (int q = 0; q < JW; ++q) glds16_(W.hi + wpan + woff[q], sb + 2 * BM * WBK + q * NWAVE * 8 * WBK);
#pragma unroll
        for (int q = 0; q < JW; ++q) glds16_(W.lo + wpan + woff[q], sb + 2 * BM * WBK + BN * WBK + q * NWAVE * 8 * WBK);
    };
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, lr = lane & 31, fsw = (lr >> 1) & 7;
    const int abase = wm * 32 * TM * WBK, wbase = 2 * BM * WBK + wn * 32 * TN * WBK;
    issue(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nkt) issue(kt + 1, (kt + 1) & 1);
        const _Float16* st = smem + (size_t)(kt & 1) * STAGE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int fo = lr * WBK + (((2 * s + h) ^ fsw) * 8);
            half8 fah[TM], fal[TM], fwh[TN], fwl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                fah[i] = *reinterpret_cast<const half8*>(st + abase + i * 32 * WBK + fo);
                fal[i] = *reinterpret_cast<const half8*>(st + abase + BM * WBK + i * 32 * WBK + fo);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                fwh[j] = *reinterpret_cast<const half8*>(st + wbase + j * 32 * WBK + fo);
                fwl[j] = *reinterpret_cast<const half8*>(st + wbase + BN * WBK + j * 32 * WBK + fo);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc1[i][j] = mfma16(fah[i], fwh[j], acc1[i][j]);
                    acc2[i][j] = mfma16(fah[i], fwl[j], acc2[i][j]);
                    acc2[i][j] = mfma16(fal[i], fwh[j], acc2[i][j]);
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 1) void k_wide(SplitPtr A, SplitPtr W, float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    int rb, cb;
    {
        const int b = blockIdx.y * gridDim.x + blockIdx.x;
        if ((gridDim.y & 7) == 0) { const int xcd = b & 7, idx = b >> 3; rb = xcd + 8 * (idx / gridDim.x); cb = idx % gridDim.x; }
        else { rb = blockIdx.y; cb = blockIdx.x; }
    }
    GemmAH ga{A, A, K, K};
    f32x16 c1[TM][TN], c2[TM][TN];
    gemm_mainloop_wide<BM, BN, WM, WN>(ga, W, M, K, rb * BM, M, cb * BN, N, smem, c1, c2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave / WN, wn = wave % WN;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb * BM + wm * 32 * TM + i * 32 + acc_row(r, lane);
                const int col = cb * BN + wn * 32 * TN + j * 32 + (lane & 31);
                C[(size_t)row * N + col] = c1[i][j][r] + c2[i][j][r] * SPLIT_INV;
            }
}

template <int BM, int BN, int TM, int TN>
__global__ __launch_bounds__(512) void k_ring(SplitPtr A, SplitPtr W, float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    int rb, cb;
    {
        const int b = blockIdx.y * gridDim.x + blockIdx.x;
        if ((gridDim.y & 7) == 0) { const int xcd = b & 7, idx = b >> 3; rb = xcd + 8 * (idx / gridDim.x); cb = idx % gridDim.x; }
        else { rb = blockIdx.y; cb = blockIdx.x; }
    }
    GemmAH ga{A, A, K, K};
    f32x16 c1[TM][TN], c2[TM][TN];
    gemm_mainloop_ring<BM, BN, TM, TN, 2>(ga, W, M, K, rb * BM, M, cb * BN, N, smem, c1, c2);
    if (threadIdx.x >= 256) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb * BM + wm * 32 * TM + i * 32 + acc_row(r, lane);
                const int col = cb * BN + wn * 32 * TN + j * 32 + (lane & 31);
                C[(size_t)row * N + col] = c1[i][j][r] + c2[i][j][r] * SPLIT_INV;
            }
}

static void split_host(float a, _Float16& hi, _Float16& lo) {
    hi = fabsf(a) < 6.103515625e-5f ? (_Float16)0.0f : (_Float16)a;
    lo = (_Float16)((a - (float)hi) * 2048.0f);
}
static void to_planes(const std::vector<float>& X, int R, int K, std::vector<_Float16>& hi, std::vector<_Float16>& lo) {
    hi.resize((size_t)R * K); lo.resize((size_t)R * K);
    for (int r = 0; r < R; ++r)
        for (int k = 0; k < K; ++k) {
            const size_t o = ((size_t)(k >> 6) * R + r) * 64 + (k & 63);
            split_host(X[(size_t)r * K + k], hi[o], lo[o]);
        }
}

template <typename F>
static float time_us(F launch, int reps = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) launch();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.0f / reps;
}

static int run_shape(int M, int N, int K) {
    std::vector<float> A((size_t)M * K), Wt((size_t)N * K);
    srand(1);
    for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 4.0f;
    for (auto& v : Wt) v = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;
    std::vector<_Float16> ah, al, wh, wl;
    to_planes(A, M, K, ah, al); to_planes(Wt, N, K, wh, wl);
    _Float16 *dah, *dal, *dwh, *dwl; float *dC, *dC2;
    CK(hipMalloc(&dah, ah.size() * 2)); CK(hipMalloc(&dal, al.size() * 2));
    CK(hipMalloc(&dwh, wh.size() * 2)); CK(hipMalloc(&dwl, wl.size() * 2));
    CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dC2, (size_t)M * N * 4));
    CK(hipMemcpy(dah, ah.data(), ah.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dal, al.data(), al.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwh, wh.data(), wh.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dwl, wl.data(), wl.size() * 2, hipMemcpyHostToDevice));
    SplitPtr sa{dah, dal}, sw{dwh, dwl};
    const double gf = 2.0 * M * N * K / 1e9;

    auto check = [&](float* dptr, const char* name) {
        std::vector<float> C((size_t)M * N);
        hipMemcpy(C.data(), dptr, C.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int s = 0; s < 4000; ++s) {
            const int r = (int)((size_t)rand() * 7919 % M), c = rand() % N;
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)A[(size_t)r * K + k] * Wt[(size_t)c * K + k];
            worst = fmax(worst, fabs(ref - C[(size_t)r * N + c]));
        }
        printf("    %-28s max |err| vs fp64 = %.3g\n", name, worst);
        return worst < 1e-4;
    };
    bool ok = true;
    auto big = [&](auto bm, auto bn, auto wm, auto wn, auto v, auto ns) {
        constexpr int BM = decltype(bm)::value, BN = decltype(bn)::value, WM = decltype(wm)::value, WN = decltype(wn)::value, V = decltype(v)::value;
        constexpr int NS = decltype(ns)::value;
        if (N % BN) return;
        const size_t lds = (size_t)NS * big_stage_halves<BM, BN>() * 2;
        hipFuncSetAttribute((const void*)k_big<BM, BN, WM, WN, V, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        dim3 grid(N / BN, M / BM);
        hipMemset(dC, 0, (size_t)M * N * 4);
        const float us = time_us([&] { hipLaunchKernelGGL((k_big<BM, BN, WM, WN, V, NS>), grid, dim3(WM * WN * 64), lds, 0, sa, sw, dC, M, N, K); });
        char name[64]; snprintf(name, sizeof name, "big %dx%d w%d v%d s%d", BM, BN, WM * WN, V, NS);
        printf("  M=%d N=%d K=%d  %-22s: %7.1f us  %6.1f TFLOP/s algorithmic (%4.1f%% of f16 peak executed x3)\n", M, N, K, name, us,
               gf / us * 1e3, 3 * gf / us * 1e3 / 2500.0 * 100);
        ok &= check(dC, name);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    using I128 = std::integral_constant<int, 128>; using I256 = std::integral_constant<int, 256>;
    using I3 = std::integral_constant<int, 3>; using I64 = std::integral_constant<int, 64>;
    big(I128{}, I256{}, I2{}, I4{}, I0{}, I3{});
    big(I128{}, I256{}, I2{}, I4{}, I1{}, I3{});
    big(I128{}, I256{}, I2{}, I4{}, I3{}, I3{});
    big(I128{}, I128{}, I2{}, I2{}, I0{}, I2{});       // 4 waves, 2-stage ring: two workgroups per CU
    big(I128{}, I128{}, I2{}, I4{}, I1{}, I3{});
    big(I64{}, I256{}, I1{}, I4{}, I0{}, I2{});
    {   // experiment: full-line k-tiles
        constexpr int BM = 128, BN = 128;
        const size_t lds = (size_t)2 * (2 * BM + 2 * BN) * 64 * 2;
        CK(hipFuncSetAttribute((const void*)k_wide<BM, BN, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        dim3 grid(N / BN, M / BM);
        hipMemset(dC, 0, (size_t)M * N * 4);
        const float us = time_us([&] { hipLaunchKernelGGL((k_wide<BM, BN, 2, 2>), grid, dim3(256), lds, 0, sa, sw, dC, M, N, K); });
        printf("  M=%d N=%d K=%d  wide 128x128 w4 k64 s2  : %7.1f us  %6.1f TFLOP/s algorithmic (%4.1f%% of f16 peak executed x3)\n", M, N, K, us,
               gf / us * 1e3, 3 * gf / us * 1e3 / 2500.0 * 100);
        ok &= check(dC, "wide 128x128 k64");
    }
    {
        constexpr int BM = 64, BN = 128;
        const size_t lds = (size_t)2 * ring_stage_halves<BM, BN>() * 2;
        CK(hipFuncSetAttribute((const void*)k_ring<BM, BN, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        dim3 grid(N / BN, M / BM);
        const float us = time_us([&] { hipLaunchKernelGGL((k_ring<BM, BN, 1, 2>), grid, dim3(512), lds, 0, sa, sw, dC2, M, N, K); });
        printf("  M=%d N=%d K=%d  ring 64x128 : %7.1f us  %6.1f TFLOP/s algorithmic\n", M, N, K, us, gf / us * 1e3);
        ok &= check(dC2, "ring 64x128");
    }
    hipFree(dah); hipFree(dal); hipFree(dwh); hipFree(dwl); hipFree(dC); hipFree(dC2);
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    int bad = 0;
    if (argc > 1) {                   // quick mode: one shape
        run_shape(32768, 256, 512);
        run_shape(32768, 512, 512);
        return 0;
    }
    for (int M : {16384, 32768}) {
        bad += run_shape(M, 768, 256);
        bad += run_shape(M, 512, 256);
        bad += run_shape(M, 512, 512);
        bad += run_shape(M, 256, 512);
    }
    printf(bad ? "FAILED\n" : "all ok\n");
    return bad;
}
