// gemm_f32.hpp - exact-fp32 matrix-core GEMM main loop shared by the LightGlue
// and ALIKED kernels (v_mfma_f32_32x32x2_f32, LDS double buffer, gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

namespace sslam {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int acc_row(int r, int lane) {  // C/D row of accumulator reg r
    return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

// ------------------------------------------------------------------------ //
//  1. GEMM  C[rows][N] = A[rows][K] . W[N][K]^T   (torch nn.Linear layout)
//     fp32 MFMA 32x32x2; block 256 threads = 2x2 waves; LDS double buffer.
// ------------------------------------------------------------------------ //
constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;   // +16 B pad: ds_read_b128 conflict-free (16 rows cover 64 banks)

template <int BM, int BN>
struct __attribute__((aligned(16))) GemmSmem {
    float a[2][BM * LDS_LD];
    float w[2][BN * LDS_LD];
};

struct GemmA {            // A operand: optional concat of two row-major sources along K
    const float* A0; int lda0;
    const float* A1; int lda1;   // lda1 == lda0 when A1 is used
    int K0;               // columns [0,K0) from A0, [K0,K) from A1 (K0 % 32 == 0)
};

#define LD4(dst, ptr) dst = *reinterpret_cast<const float4*>(ptr)
#define ST4(ptr, src) *reinterpret_cast<float4*>(ptr) = src

template <int BM, int BN, int TM, int TN>
__device__ __forceinline__ void gemm_mainloop(const GemmA& ga, const float* __restrict__ W, int ldw,
                                              int K, int row0, int row_cap, int col0, int col_cap,
                                              GemmSmem<BM, BN>& sm, f32x16 (&acc)[TM][TN]) {
    static_assert(BM == 64 * TM && BN == 64 * TN, "2x2 waves of 32*TM x 32*TN");
    constexpr int NA = BM / 32, NW = BN / 32;      // float4 per thread per k-tile
    static_assert(NA <= 4 && NW <= 4, "named prefetch registers cover up to 128-row tiles");
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, lr = lane & 31;

    // Register prefetch of the next k-tile in NAMED registers (an indexed float4 array here
    // ends up as a private array in scratch / LDS and exposes the whole load latency).
    float4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
    ra0 = ra1 = ra2 = ra3 = rw0 = rw1 = rw2 = rw3 = make_float4(0, 0, 0, 0);
    const int lr8 = t >> 3, lc4 = (t & 7) * 4;     // this thread's (row, k-offset) in a 32-row slab
    size_t oa[4], ow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        oa[j] = (size_t)min(row0 + lr8 + 32 * j, row_cap - 1) * ga.lda0 + lc4;
        ow[j] = (size_t)min(col0 + lr8 + 32 * j, col_cap - 1) * ldw + lc4;
    }
    const int sto = lr8 * LDS_LD + lc4;

#define GEMM_GLOAD(kt_)                                                        \
    {                                                                          \
        const int k_ = (kt_) * BK;                                             \
        const float* pa_ = (k_ < ga.K0) ? ga.A0 + k_ : ga.A1 + (k_ - ga.K0);   \
        const float* pw_ = W + k_;                                             \
        LD4(ra0, pa_ + oa[0]);                                                 \
        if constexpr (NA > 1) LD4(ra1, pa_ + oa[1]);                           \
        if constexpr (NA > 2) LD4(ra2, pa_ + oa[2]);                           \
        if constexpr (NA > 3) LD4(ra3, pa_ + oa[3]);                           \
        LD4(rw0, pw_ + ow[0]);                                                 \
        if constexpr (NW > 1) LD4(rw1, pw_ + ow[1]);                           \
        if constexpr (NW > 2) LD4(rw2, pw_ + ow[2]);                           \
        if constexpr (NW > 3) LD4(rw3, pw_ + ow[3]);                           \
    }
#define GEMM_SSTORE(buf_)                                                      \
    {                                                                          \
        float* da_ = &sm.a[buf_][sto];                                         \
        float* dw_ = &sm.w[buf_][sto];                                         \
        ST4(da_, ra0);                                                         \
        if constexpr (NA > 1) ST4(da_ + 32 * LDS_LD, ra1);                     \
        if constexpr (NA > 2) ST4(da_ + 64 * LDS_LD, ra2);                     \
        if constexpr (NA > 3) ST4(da_ + 96 * LDS_LD, ra3);                     \
        ST4(dw_, rw0);                                                         \
        if constexpr (NW > 1) ST4(dw_ + 32 * LDS_LD, rw1);                     \
        if constexpr (NW > 2) ST4(dw_ + 64 * LDS_LD, rw2);                     \
        if constexpr (NW > 3) ST4(dw_ + 96 * LDS_LD, rw3);                     \
    }

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nkt = K / BK;
    GEMM_GLOAD(0);
    GEMM_SSTORE(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) GEMM_GLOAD(kt + 1);
        const float* sa = sm.a[cur] + (wm * 32 * TM + lr) * LDS_LD + 4 * h;
        const float* sw = sm.w[cur] + (wn * 32 * TN + lr) * LDS_LD + 4 * h;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            float4 af[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) LD4(af[i], sa + i * 32 * LDS_LD + g * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) LD4(wf[j], sw + j * 32 * LDS_LD + g * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = mfma32(af[i].x, wf[j].x, acc[i][j]);
                    acc[i][j] = mfma32(af[i].y, wf[j].y, acc[i][j]);
                    acc[i][j] = mfma32(af[i].z, wf[j].z, acc[i][j]);
                    acc[i][j] = mfma32(af[i].w, wf[j].w, acc[i][j]);
                }
        }
        if (kt + 1 < nkt) GEMM_SSTORE(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#undef GEMM_GLOAD
#undef GEMM_SSTORE
}


}  // namespace sslam
