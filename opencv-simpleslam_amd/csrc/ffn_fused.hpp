// ffn_fused.hpp - the whole FFN of a LightGlue transformer block as ONE kernel (gfx950, split precision).
//
//     x += W2 . GELU(LayerNorm(W1 . [x | message] + b1)) + b2          (oracle/lightglue_ref.py:73-77 `_ffn`;
//                                                                       upstream TransformerLayer.ffn / CrossBlock.ffn)
//
// r02 ran this as two launches (a 64 x 512 whole-row GEMM with LayerNorm + GELU in its epilogue, then a
// 128 x 128 GEMM with the residual): the 512-wide hidden went out to HBM as fp16 plane pairs (67 MB per
// 8-pair launch) and came back, and the first kernel's lock-step main loop ran at a third of the rate
// the LDS-DMA path of a CU sustains.  Here a workgroup (8 waves, one per CU) owns 64 tokens end to end
// and the hidden never leaves its registers:
//
//   phase 1   h^T[j][tok] = W1[j][:] . a[tok][:]       the TRANSPOSED product: W1 rows are the MFMA's
//             A operand, token rows its B operand, so the accumulator tile has the token on the LANE
//             and the hidden index j in the REGISTERS.  Wave (wj, wt) owns j in [128 wj, +128) x tokens
//             [32 wt, +32): 4 tiles of 32 x 32, two fp32 accumulators each (hi.hi and the cross terms).
//             W1 / a k-tiles (32 deep, 72 KB) stream through a 2-stage LDS ring by LDS-DMA; the k-loop
//             is software-pipelined with ONE barrier per k-tile, placed between its two k16 steps:
//             at that point every fragment of the tile is in registers (its stage is free for tile
//             kt+2) and the wave has waited for its own pieces of tile kt+1.
//   LN/GELU   LayerNorm statistics of a token are sums over the registers of its lane (+ the other lane
//             half + the 4 wj waves through 2 KB of LDS); GELU and the hi/lo split run in registers.
//   phase 2   y^T[n][tok] = W2[n][:] . g[tok][:]       sums over j = the ROW index of the phase-1 tile, so
//             the split hidden IS the B operand of these MFMAs as it stands (no LDS, no lane movement;
//             the k order inside a 16-step is the accumulator's row order, and W2's columns are stored
//             in that order at pack time).  Each wave holds only its own 128 j: it multiplies the matching
//             W2 columns (K split over the 4 wj waves) for 64 output columns at a time (4 quarters, 64
//             accumulator registers), the four partial tiles are added through LDS in a fixed order,
//             and the quarter's epilogue (+ b2, + residual, fp32 state and its split planes, whole rows
//             of 256 B) runs on all 8 waves.  W2 streams through a 3-stage ring of 32 KB stages that
//             are single contiguous, pre-swizzled blocks in HBM (`ffn_w2_fused_index`).
//
// DMA intake per workgroup: 1.18 MB (W1 + a) + 0.5 MB (W2) for 151 M executed MFMA flop-pairs: the kernel
// is bound by the LDS-DMA path (~46 B/clk/CU from L2), not by the matrix pipe.
#pragma once
#include "gemm_f16x3.hpp"

namespace sslam {

constexpr int FFN_TOK = 64;                    // tokens per workgroup
constexpr int FFN_D = 256, FFN_H = 512;        // model width, hidden width
constexpr int FFN_P1_STAGE = (2 * FFN_H + 2 * FFN_TOK) * 32;          // halves per phase-1 stage (73 728 B)
constexpr int FFN_P2_STAGE = 4 * 2 * 64 * 32;                         // halves per phase-2 stage (32 768 B)
constexpr int FFN_CONST_OFF = 2 * FFN_P1_STAGE * 2;                   // bytes: b1 | ln_w | ln_b (fp32)
constexpr int FFN_RED_OFF = FFN_CONST_OFF + 3 * FFN_H * 4;            // bytes: LayerNorm partial sums [2][4][64]
constexpr int FFN_LDS_BYTES = FFN_RED_OFF + 2 * 4 * FFN_TOK * 4;      // 155 648
constexpr int FFN_P2_RED_OFF = 3 * FFN_P2_STAGE * 2;                  // bytes: K-split partial tiles (32 KB)
constexpr int FFN_P2_Y_OFF = FFN_P2_RED_OFF + 32768;                  // bytes: one quarter of y, fp32 [64][68]
constexpr int FFN_Y_LD = 68;
static_assert(FFN_P2_Y_OFF + FFN_TOK * FFN_Y_LD * 4 <= FFN_LDS_BYTES, "phase-2 regions fit (the phase-1 constants are dead by then)");
static_assert(2 * FFN_P2_STAGE * 2 <= FFN_P1_STAGE * 2, "the first two W2 stages lie inside phase-1 stage 0");

// accumulator row -> position inside a 16-deep k step: element e (0..7) of lane half h of the B fragment
// built from accumulator registers 8 s .. 8 s + 7 is row 16 s + (e & 3) + 8 (e >> 2) + 4 h of the tile;
// the MFMA calls that slot k = 8 h + e.  W2's column for slot k of a 16-group:
__host__ __device__ inline int ffn_kperm(int k) { return (k & 3) + 8 * ((k >> 2) & 1) + 4 * (k >> 3); }

// Fused-stage layout of W2 [256][512]: stage (nq, jt) = 32 KB = [wj 4][plane 2][n 64][32 halves], where
// n = 64 nq + nl, slot kk of the row holds column j = 128 wj + 32 jt + 16 (kk >> 4) + ffn_kperm(kk & 15),
// and the 16-byte chunk c of a row sits at chunk position c ^ ((nl >> 2) & 3) (the LDS read swizzle,
// applied here so the DMA is a linear copy).  Returns the offset in halves of (plane, n, j).
__host__ __device__ inline size_t ffn_w2_fused_index(int plane, int n, int j) {
    const int nq = n >> 6, nl = n & 63, wj = j >> 7, jt = (j >> 5) & 3, jl = j & 31;
    const int g16 = jl >> 4, r = jl & 15;
    // inverse of ffn_kperm on 0..15: row r = (k & 3) + 8 ((k >> 2) & 1) + 4 (k >> 3)
    const int k = (r & 3) + 4 * ((r >> 3) & 1) + 8 * ((r >> 2) & 1);
    const int kk = 16 * g16 + k;
    const int chunk = (kk >> 3) ^ ((nl >> 2) & 3);
    return ((((size_t)(nq * 4 + jt) * 4 + wj) * 2 + plane) * 64 + nl) * 32 + chunk * 8 + (kk & 7);
}

struct FfnFusedArgs {
    SplitPtr xs, msgs;          // token state / attention context planes, k-panel layout (PANEL_K = 32) over plane_rows rows
    int plane_rows;
    SplitPtr w1;                // W1 [512][512] planes, k-panel layout over 512 rows
    const float* b1; const float* ln_w; const float* ln_b;
    const _Float16* w2f;        // W2, both planes, in the fused-stage layout (ffn_w2_fused_index)
    const float* b2;
    float* x;                   // fp32 token state [plane_rows][256]: residual in, new state out
    _Float16* xo_hi; _Float16* xo_lo;   // split planes of the new state (k-panel layout; may alias xs)
    unsigned long long* stamps;         // diagnostic builds only (FFN_STAMP): [workgroup][8] s_memtime values; else unused
};

#ifdef FFN_STAMP
#define FFN_STAMP_AT(i_) do { if (threadIdx.x == 0) { unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); p.stamps[blockIdx.x * 8 + (i_)] = t_; } } while (0)
#else
#define FFN_STAMP_AT(i_)
#endif
#ifndef FFN_ABL
#define FFN_ABL 0       // ubench ablations: 1 no MFMA, 2 no DMA after the prologue, 4 no LN/GELU arithmetic, 8 no stores
#endif

// Instruction order of one software-pipelined block: 12 MFMAs on the fragments in registers, with the
// NR ds_read_b128 of the next fragments and the ND LDS-DMA pieces of a later tile placed one per MFMA
// behind the first ones (a wave then never sits in a run of reads or DMA issues while the matrix pipe idles).
// FFN_SCHED: 0 = scheduler's choice, 1 = sched_group_barrier recipe, 2 = reads / DMA first, then the MFMAs (hard pin).
#ifndef FFN_SCHED
#define FFN_SCHED 1
#endif
#if FFN_SCHED == 1
#define FFN_PIN_READS_UNDER_MFMAS(NR, ND)                                          \
    _Pragma("unroll") for (int ig_ = 0; ig_ < 12; ++ig_) {                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                         \
        if (ig_ < (ND)) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);         \
        if (ig_ < (NR)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         \
    }
#elif FFN_SCHED == 2
#define FFN_PIN_READS_UNDER_MFMAS(NR, ND)                                          \
    __builtin_amdgcn_sched_group_barrier(0x010, (ND), 0);                          \
    __builtin_amdgcn_sched_group_barrier(0x100, (NR), 0);                          \
    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
#else
#define FFN_PIN_READS_UNDER_MFMAS(NR, ND)
#endif

__device__ __forceinline__ auto ffn_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// one LDS-DMA wave-instruction: 64 lanes x 16 B from (descriptor base + scalar offset + lane offset) to lds .. + 1 KiB
template <typename R>
__device__ __forceinline__ void ffn_dma16(R rsrc, int voff_bytes, int soff_bytes, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff_bytes,
                                             soff_bytes, 0, 0);
}

__device__ __forceinline__ float ffn_erf(float x) {        // Abramowitz & Stegun 7.1.26 (see lightglue_kernels.hip erf_as)
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
    return copysignf(fmaf(-p * t, e, 1.0f), x);
}

// One 64-token tile.  grow0: plane row of token 0; grow_cap: one past the last plane row that may be read
// (rows are clamped to it); n_valid: tokens of the tile that exist (stores are masked to them).
// 512 threads; `smem` = FFN_LDS_BYTES of dynamic LDS, 16-byte aligned.
__device__ __forceinline__ void ffn_fused_tile(const FfnFusedArgs& p, int grow0, int grow_cap, int n_valid,
                                               int* range_flag, _Float16* smem) {
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wj = wave >> 1, wt = wave & 1;
    const int h = lane >> 5, lr = lane & 31;
    char* const smem_b = reinterpret_cast<char*>(smem);
    float* const cst = reinterpret_cast<float*>(smem_b + FFN_CONST_OFF);        // b1 | ln_w | ln_b
    float* const red = reinterpret_cast<float*>(smem_b + FFN_RED_OFF);

    FFN_STAMP_AT(0);
    // constants of the LayerNorm epilogue into LDS (ordinary loads: before any DMA is in flight)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float* src = i == 0 ? p.b1 : (i == 1 ? p.ln_w : p.ln_b);
        cst[i * FFN_H + t] = src[t];
    }

    // ------------------------------------------------------------------ phase 1: h^T = W1 . a^T
    f32x16 c1[4], c2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { c1[i][r] = 0.0f; c2[i][r] = 0.0f; }

    // producer side: wave w issues W pieces w, w + 8, w + 16, w + 24 of each plane (16 rows x 64 B each)
    // and one piece of the token tile (waves 0-3: hi plane, 4-7: lo plane).  LDS-DMA in its BUFFER form
    // (buffer_load_dwordx4 ... lds): the plane's base sits in a scalar descriptor, the k-tile advance in a
    // scalar offset and only the loop-invariant 32-bit lane offset in a VGPR - no 64-bit address arithmetic
    // or address registers per piece (the flat form cost ~25 VGPRs of hoisted addresses here).
    const int prow = lane >> 2, pc = lane & 3;
    const int psw = (pc ^ ((prow >> 2) & 3)) * 8;
    int woff[4];                                   // bytes
#pragma unroll
    for (int q = 0; q < 4; ++q) woff[q] = (((wave + 8 * q) * 16 + prow) * PANEL_K + psw) * 2;
    const int aoff = (min(grow0 + (wave & 3) * 16 + prow, grow_cap - 1) * PANEL_K + psw) * 2;
    const bool a_lo = wave >= 4;
    const auto r_w1h = ffn_rsrc(p.w1.hi, FFN_H * FFN_H * 2), r_w1l = ffn_rsrc(p.w1.lo, FFN_H * FFN_H * 2);
    const unsigned a_bytes = (unsigned)p.plane_rows * FFN_D * 2;
    const auto r_ax = ffn_rsrc(a_lo ? p.xs.lo : p.xs.hi, a_bytes), r_am = ffn_rsrc(a_lo ? p.msgs.lo : p.msgs.hi, a_bytes);
    const auto r_w2 = ffn_rsrc(p.w2f, 2 * FFN_D * FFN_H * 2);
    auto issue1 = [&](int kt, int stage) {
#if FFN_ABL & 2
        if (kt > 1) return;
#endif
        const int wpan = kt * (FFN_H * PANEL_K * 2);                 // bytes
        _Float16* st = smem + (size_t)stage * FFN_P1_STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) ffn_dma16(r_w1h, woff[q], wpan, st + (wave + 8 * q) * 16 * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) ffn_dma16(r_w1l, woff[q], wpan, st + FFN_H * 32 + (wave + 8 * q) * 16 * 32);
        _Float16* ad = st + 2 * FFN_H * 32 + (a_lo ? FFN_TOK * 32 : 0) + (wave & 3) * 16 * 32;
        if (kt < 8) ffn_dma16(r_ax, aoff, kt * p.plane_rows * (PANEL_K * 2), ad);
        else ffn_dma16(r_am, aoff, (kt - 8) * p.plane_rows * (PANEL_K * 2), ad);
    };
    // phase-2 producer: stage block `step` (32 KB contiguous) -> ring stage step % 3; 4 pieces per wave
    auto issue2 = [&](int step) {
#if FFN_ABL & 2
        if (step > 2) return;
#endif
        _Float16* st = smem + (size_t)(step % 3) * FFN_P2_STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int piece = wave + 8 * q;                         // 32 pieces of 1 KiB: the block is copied as it stands
            ffn_dma16(r_w2, lane * 16, step * (FFN_P2_STAGE * 2) + piece * 1024, st + piece * 512);
        }
    };

    // consumer side: fragment offsets (halves) inside a stage
    const int fsw = (lr >> 2) & 3;
    const int fo0 = ((0 + h) ^ fsw) * 8, fo1 = ((2 + h) ^ fsw) * 8;            // k16 step 0 / 1
    const int wrow = (128 * wj + lr) * 32, arow = 2 * FFN_H * 32 + (32 * wt + lr) * 32;
    auto read1 = [&](const _Float16* st, int fo, half8 (&wh)[4], half8 (&wl)[4], half8& ah, half8& al) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wh[i] = *reinterpret_cast<const half8*>(st + wrow + i * 32 * 32 + fo);
            wl[i] = *reinterpret_cast<const half8*>(st + FFN_H * 32 + wrow + i * 32 * 32 + fo);
        }
        ah = *reinterpret_cast<const half8*>(st + arow + fo);
        al = *reinterpret_cast<const half8*>(st + arow + FFN_TOK * 32 + fo);
    };
    auto mma1 = [&](const half8 (&wh)[4], const half8 (&wl)[4], const half8& ah, const half8& al) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#if FFN_ABL & 1
            c1[i][0] += (float)wh[i][0] + (float)ah[1];
            c2[i][0] += (float)wl[i][0] + (float)al[1];
#else
            c1[i] = mfma16(wh[i], ah, c1[i]);
            c2[i] = mfma16(wh[i], al, c2[i]);
            c2[i] = mfma16(wl[i], ah, c2[i]);
#endif
        }
    };

    constexpr int NKT = 2 * FFN_D / 32;          // 16 k-tiles
    half8 wh0[4], wl0[4], ah0, al0, wh1[4], wl1[4], ah1, al1;
    issue1(0, 0);
    issue1(1, 1);
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");           // own pieces of tile 0 (tile 1's nine stay in flight)
    __builtin_amdgcn_s_barrier();
    read1(smem, fo0, wh0, wl0, ah0, al0);
    FFN_STAMP_AT(1);
    for (int kt = 0; kt < NKT; ++kt) {
        const _Float16* st = smem + (size_t)(kt & 1) * FFN_P1_STAGE;
        // k16 step 0 multiplies while the fragments of step 1 are read: left to itself the scheduler
        // sinks every read to just in front of its consumer (2-4 MFMAs per exposed LDS round trip)
        read1(st, fo1, wh1, wl1, ah1, al1);
        mma1(wh0, wl0, ah0, al0);
        FFN_PIN_READS_UNDER_MFMAS(10, 0);
        if (kt + 1 < NKT) {
            // own pieces of tile kt+1 landed; every fragment of tile kt is in registers
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < NKT) issue1(kt + 2, kt & 1);
            else { issue2(0); issue2(1); }                      // kt == 14: phase-1 stage 0 is free - the first two W2 stages
            read1(smem + (size_t)((kt + 1) & 1) * FFN_P1_STAGE, fo0, wh0, wl0, ah0, al0);
            mma1(wh1, wl1, ah1, al1);
            FFN_PIN_READS_UNDER_MFMAS(10, 9);
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mma1(wh1, wl1, ah1, al1);
        }
    }

    FFN_STAMP_AT(2);
    // ------------------------------------------------------------------ LayerNorm + GELU + split, in registers
    // v[i][r] = h[j = 128 wj + 32 i + (r & 3) + 8 (r >> 2) + 4 h][tok = 32 wt + lr]
    float v[4][16];
    {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = *reinterpret_cast<const float4*>(cst + 128 * wj + 32 * i + 8 * g + 4 * h);
                const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    v[i][r] = (c1[i][r] + c2[i][r] * SPLIT_INV) + bb[e];
                    s += v[i][r];
                }
            }
        s += __shfl_xor(s, 32);
        if (h == 0) red[wj * FFN_TOK + 32 * wt + lr] = s;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int tk = 32 * wt + lr;
        const float mean = ((red[tk] + red[FFN_TOK + tk]) + (red[2 * FFN_TOK + tk] + red[3 * FFN_TOK + tk])) / 512.0f;
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { v[i][r] -= mean; q += v[i][r] * v[i][r]; }
        q += __shfl_xor(q, 32);
        float* red2 = red + 4 * FFN_TOK;
        if (h == 0) red2[wj * FFN_TOK + tk] = q;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const float var = ((red2[tk] + red2[FFN_TOK + tk]) + (red2[2 * FFN_TOK + tk] + red2[3 * FFN_TOK + tk])) / 512.0f;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[i][r] *= rstd;
    }
    // g = GELU(v * gamma + beta) -> B fragments of phase 2: gh[i][s] / gl[i][s] = registers 8 s .. 8 s + 7 of tile i
    half8 gh[4][2], gl[4][2];
    float amax = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int jb = 128 * wj + 32 * i + 8 * g + 4 * h;
            const float4 gm = *reinterpret_cast<const float4*>(cst + FFN_H + jb);
            const float4 bt = *reinterpret_cast<const float4*>(cst + 2 * FFN_H + jb);
            const float gmm[4] = {gm.x, gm.y, gm.z, gm.w}, btt[4] = {bt.x, bt.y, bt.z, bt.w};
            float ge[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = v[i][4 * g + e] * gmm[e] + btt[e];
#if FFN_ABL & 4
                ge[e] = y;
#else
                ge[e] = 0.5f * y * (1.0f + ffn_erf(y * 0.70710678118654752440f));
#endif
            }
            unsigned h01, l01, h23, l23;
            split2_fast(ge[0], ge[1], h01, l01, amax);
            split2_fast(ge[2], ge[3], h23, l23, amax);
            // registers 4 g .. 4 g + 3 = elements 4 (g & 1) .. + 3 of k16 step s = g >> 1
            typedef unsigned uint2v __attribute__((ext_vector_type(2)));
            const uint2v hw = {h01, h23}, lw = {l01, l23};
            const half4 h4 = __builtin_bit_cast(half4, hw), l4 = __builtin_bit_cast(half4, lw);
#pragma unroll
            for (int e = 0; e < 4; ++e) { gh[i][g >> 1][4 * (g & 1) + e] = h4[e]; gl[i][g >> 1][4 * (g & 1) + e] = l4[e]; }
        }

    FFN_STAMP_AT(3);
    // ------------------------------------------------------------------ phase 2: y^T = W2 . g^T (K split over wj)
    // every wave is past its last read of the phase-1 ring and of the LayerNorm scratch
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue2(2);
    // fragment offsets inside a phase-2 stage: [wj][plane][64 n][32], chunk position pre-swizzled with (n >> 2) & 3
    const int w2row = (wj * 2 * 64 + lr) * 32;
    float* const pred = reinterpret_cast<float*>(smem_b + FFN_P2_RED_OFF);
    float* const ybuf = reinterpret_cast<float*>(smem_b + FFN_P2_Y_OFF);
    auto read2 = [&](int step, half8 (&ah)[2][2], half8 (&al)[2][2]) {          // [n tile][k16 step]
        const _Float16* st = smem + (size_t)(step % 3) * FFN_P2_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int fo = (((2 * s + h) ^ fsw) * 8);
                ah[i][s] = *reinterpret_cast<const half8*>(st + w2row + i * 32 * 32 + fo);
                al[i][s] = *reinterpret_cast<const half8*>(st + w2row + 64 * 32 + i * 32 * 32 + fo);
            }
    };
    f32x16 d1[2], d2[2];
    half8 pa[2][2], pl[2][2], qa[2][2], ql[2][2];          // fragment sets of even / odd steps
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // own pieces of step 0 (steps 1, 2 stay in flight)
    __builtin_amdgcn_s_barrier();
    read2(0, pa, pl);
    FFN_STAMP_AT(4);
    for (int nq = 0; nq < 4; ++nq) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { d1[i][r] = 0.0f; d2[i][r] = 0.0f; }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const int step = nq * 4 + jt;
            // step + 1: own pieces landed (issued so far: steps <= step + 2), every fragment of `step` is in registers
            if (step + 1 < 16) {
                if (step + 2 < 16) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();        // step + 1 visible to all; the stage of `step` is free
                if (step + 3 < 16) issue2(step + 3);
                if (jt & 1) read2(step + 1, pa, pl); else read2(step + 1, qa, ql);
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const half8 fa = (jt & 1) ? qa[i][s] : pa[i][s], fl = (jt & 1) ? ql[i][s] : pl[i][s];
#if FFN_ABL & 1
                    d1[i][0] += (float)fa[0] + (float)gh[jt][s][1];
                    d2[i][0] += (float)fl[0] + (float)gl[jt][s][1];
#else
                    d1[i] = mfma16(fa, gh[jt][s], d1[i]);
                    d2[i] = mfma16(fa, gl[jt][s], d2[i]);
                    d2[i] = mfma16(fl, gh[jt][s], d2[i]);
#endif
                }
            if (step + 1 < 16) { FFN_PIN_READS_UNDER_MFMAS(8, 4); }
        }

        if (nq == 0) FFN_STAMP_AT(5);
        // ---- quarter nq complete in every wave: add the four K-slices ((p0 + p1) + (p2 + p3)), fixed order
        float y[2][16];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[i][r] = d1[i][r] + d2[i][r] * SPLIT_INV;
        float4* const slot = reinterpret_cast<float4*>(pred) + (size_t)(((wj >> 1) * 2 + wt) * 8) * 64 + lane;
        if (wj & 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    slot[(i * 4 + g) * 64] = make_float4(y[i][4 * g], y[i][4 * g + 1], y[i][4 * g + 2], y[i][4 * g + 3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!(wj & 1)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 o = slot[(i * 4 + g) * 64];
                    y[i][4 * g] += o.x; y[i][4 * g + 1] += o.y; y[i][4 * g + 2] += o.z; y[i][4 * g + 3] += o.w;
                }
            if (wj == 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        slot[(i * 4 + g) * 64] = make_float4(y[i][4 * g], y[i][4 * g + 1], y[i][4 * g + 2], y[i][4 * g + 3]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wj == 0) {
            const float4* const s1 = reinterpret_cast<const float4*>(pred) + (size_t)((2 + wt) * 8) * 64 + lane;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 o = s1[(i * 4 + g) * 64];
                    // tile rows 8 g + 4 h .. + 3 of n tile i, token 32 wt + lr
                    *reinterpret_cast<float4*>(ybuf + (32 * wt + lr) * FFN_Y_LD + 32 * i + 8 * g + 4 * h) =
                        make_float4(y[i][4 * g] + o.x, y[i][4 * g + 1] + o.y, y[i][4 * g + 2] + o.z, y[i][4 * g + 3] + o.w);
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (nq == 0) FFN_STAMP_AT(6);
        {   // quarter epilogue on all 512 threads: thread = (token t >> 3, 8 columns 8 (t & 7) of the quarter)
            const int tok = t >> 3, cl = (t & 7) * 8, col = 64 * nq + cl;
            if (tok < n_valid) {
                const float4 ya = *reinterpret_cast<const float4*>(ybuf + tok * FFN_Y_LD + cl);
                const float4 yb = *reinterpret_cast<const float4*>(ybuf + tok * FFN_Y_LD + cl + 4);
                const float4 ba = *reinterpret_cast<const float4*>(p.b2 + col), bb = *reinterpret_cast<const float4*>(p.b2 + col + 4);
                float* xr = p.x + (size_t)(grow0 + tok) * FFN_D + col;
                const float4 xa = *reinterpret_cast<const float4*>(xr), xb = *reinterpret_cast<const float4*>(xr + 4);
                float o[8] = {(ya.x + ba.x) + xa.x, (ya.y + ba.y) + xa.y, (ya.z + ba.z) + xa.z, (ya.w + ba.w) + xa.w,
                              (yb.x + bb.x) + xb.x, (yb.y + bb.y) + xb.y, (yb.z + bb.z) + xb.z, (yb.w + bb.w) + xb.w};
#if !(FFN_ABL & 8)
                *reinterpret_cast<float4*>(xr) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(xr + 4) = make_float4(o[4], o[5], o[6], o[7]);
                uint4 hi, lo;
                split8_fast(o, hi, lo, amax);
                const size_t po = panel_index(grow0 + tok, col, p.plane_rows);
                *reinterpret_cast<uint4*>(p.xo_hi + po) = hi;
                *reinterpret_cast<uint4*>(p.xo_lo + po) = lo;
#else
                if (o[0] == 123.456f) *xr = o[1];
#endif
            }
        }
    }
    split_range_check(amax, range_flag);
    FFN_STAMP_AT(7);
}

}  // namespace sslam
