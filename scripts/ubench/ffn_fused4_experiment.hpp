// ffn_fused4_experiment.hpp (r04, NOT part of the product: measured slower, see below; it was wired into
// csrc/lightglue_kernels.hip as lg_ffn_fused4_kernel / debug_big_gemm(lg, 6) at commit a862be4, scripts/ubench/ffn4_check.py and
// ab_ffn4.sh ran against that build) - the fused FFN (ffn_fused.hpp) as 32-token tiles on FOUR waves with a compact LDS image, so that TWO
// workgroups share a CU (r04).
//
// The 64-token / 8-wave kernel keeps one workgroup per CU (135 KB of LDS, 230 registers x 2 waves per SIMD): its matrix
// phases (37 k of 77 k cycles per tile) and its vector phases (operand prologue, LayerNorm / GELU, epilogue) run one after
// the other with nothing beside them.  Here a workgroup is half the tokens on half the waves - each wave owns TWO of the
// eight 64-column slots of the hidden layer (and two of the eight 32-column slots of the output), so a wave does the work
// of one 8-wave wave on 64 tokens - and needs 75 KB of LDS: two workgroups per CU, one's LayerNorm / GELU under the other's
// MFMAs.  Price: every weight byte crosses L2 -> CU once per 32 tokens instead of once per 64.
// Arithmetic, fragment layouts of W1 / W2 and every summation order are those of ffn_fused_tile (the LayerNorm partial sums
// stay per 64-column slot and are added in the same tree): results are bit-identical, tested (matches, scores, token states,
// early stop and pruning on ragged batches: ffn4_check.py).
// MEASURED (8 pairs of 2048 x 2048 per launch, kernel trace, one box): 99.2 us against 91 - 93 us for the 64-token / 8-wave kernel;
// deeper weight rings spill (FFN4_D2 = 8: 113 us, FFN4_D1 = 4: 178 us).  Two co-resident half-size workgroups do overlap their
// vector and matrix phases, but every workgroup streams all 1.5 MB of W1 + W2 through the CU for 32 tokens: the doubled weight
// stream costs more than the overlap returns.  VERDICT r03 item 6, option 1: closed by measurement.
#pragma once
#include "ffn_fused.hpp"

namespace sslam {

constexpr int FFN4_TOK = 32;
constexpr int FFN4_OPER_BYTES = 65536;         // operand tile [16 k-panels][2 planes][32 tok][32] = hidden fragments [2][32 ks][64 lanes][8]
constexpr int FFN4_RED_OFF = FFN4_OPER_BYTES;  // LayerNorm partial sums [2 passes][8 slots][32 tok] fp32
constexpr int FFN4_CONST_OFF = FFN4_RED_OFF + 2 * 8 * FFN4_TOK * 4;
constexpr int FFN4_LDS_BYTES = FFN4_CONST_OFF + (3 * FFN_H + 3 * FFN_D) * 4;      // 76 800: two workgroups per CU
static_assert(FFN4_TOK * FFN_Y_LD * 4 <= FFN4_OPER_BYTES, "the output tile is staged where the hidden fragments were");
static_assert(2 * FFN4_LDS_BYTES <= 160 * 1024, "two workgroups per CU");

#ifndef FFN4_D1
#define FFN4_D1 2       // W1 fragment sets in flight per wave (steps of 8 KB: two slots)
#endif
#ifndef FFN4_D2
#define FFN4_D2 4       // W2 fragment sets in flight per wave (steps of 4 KB: two slots)
#endif
constexpr int FFN4_LEADC = 2;
// vector-memory operations a wave issues after the last piece of chunk c and before chunk c must be in place (see
// ffn_ops_after_chunk: the same replay with 8 W1 loads per step and FFN4_D1 sets)
constexpr int ffn4_ops_after_chunk(int c) {
    int n = 0; bool seen = false;
    for (int k = 0; k < FFN4_LEADC; ++k) {
        if (seen) n += 2;
        if (k == c) seen = true;
        if (k == 0 && seen) n += 8 * (FFN4_D1 - 1);
    }
    if (c == 0) return n;
    for (int s = 0; s < 32; ++s) {
        if (s == 4 * c - 1) return n;
        if (s + FFN4_D1 < 32 && seen) n += 8;
        if (s % 4 == 0 && s / 4 + FFN4_LEADC < 8) { if (seen) n += 2; if (s / 4 + FFN4_LEADC == c) seen = true; }
    }
    return n;
}

// One tile of 32 tokens.  grow0 / grow_cap / n_valid / range_flag as in ffn_fused_tile.  256 threads; `smem` =
// FFN4_LDS_BYTES of dynamic LDS, 16-byte aligned.
__device__ __forceinline__ void ffn_fused_tile4(const FfnFusedArgs& p, int grow0, int grow_cap, int n_valid,
                                                int* range_flag, _Float16* smem) {
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);          // 0..3: hidden slots 2 wave, 2 wave + 1
    const int h = lane >> 5, lr = lane & 31;
    char* const smem_b = reinterpret_cast<char*>(smem);
    float* const red = reinterpret_cast<float*>(smem_b + FFN4_RED_OFF);
    const int lane16 = lane * 16;
    float* const cst = reinterpret_cast<float*>(smem_b + FFN4_CONST_OFF);
    {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = t + 256 * q;
            cst[i] = p.b1[i]; cst[FFN_H + i] = p.ln_w[i]; cst[2 * FFN_H + i] = p.ln_b[i];
        }
        cst[3 * FFN_H + t] = p.b2[t];
        if (p.hm != nullptr) {
            cst[3 * FFN_H + FFN_D + t] = p.hm[t];
            cst[3 * FFN_H + 2 * FFN_D + t] = p.hc != nullptr ? p.hc[t] : 0.0f;
        }
    }

    const auto r_w1 = ffn_rsrc(p.w1f, 2 * FFN_H * FFN_H * 2), r_w2 = ffn_rsrc(p.w2f, 2 * FFN_D * FFN_H * 2);
    // W1 fragment set of step ks for this wave's two slots: [slot][hi jt0, hi jt1, lo jt0, lo jt1], 1 KiB each
    auto load_w1 = [&](int ks, half8 (&dst)[8]) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int base = (ks * 8 + 2 * wave + sl) * 4096;
#pragma unroll
            for (int f = 0; f < 4; ++f) dst[4 * sl + f] = ffn_ldfrag(r_w1, lane16, base + f * 1024);
        }
    };
    auto load_w2 = [&](int ks, half8 (&dst)[4]) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int base = (ks * 8 + 2 * wave + sl) * 2048;
            dst[2 * sl] = ffn_ldfrag(r_w2, lane16, base);
            dst[2 * sl + 1] = ffn_ldfrag(r_w2, lane16, base + 1024);
        }
    };

    // ------------------------------------------------------------------ prologue
    // operand tile by LDS-DMA: wave w brings rows 16 (w & 1) .. + 15 of plane (w >> 1) of every k-panel
    half8 wq[FFN4_D1][8];
    load_w1(0, wq[0]);
    const int prow = lane >> 2, pc = lane & 3;
    const int psw = (pc ^ ((prow >> 2) & 3)) * 8;
    const int aoff = (min(grow0 + (wave & 1) * 16 + prow, grow_cap - 1) * PANEL_K + psw) * 2;      // bytes
    const bool lo = wave >= 2;
    const unsigned a_bytes = (unsigned)p.plane_rows * FFN_D * 2;
    const auto r_ax = ffn_rsrc(lo ? p.xs.lo : p.xs.hi, a_bytes), r_am = ffn_rsrc(lo ? p.msgs.lo : p.msgs.hi, a_bytes);
    const int pstride = p.plane_rows * (PANEL_K * 2);               // bytes per k-panel of a plane
    auto issue_chunk = [&](int c) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kp = 2 * c + j;
            _Float16* dst = smem + ((kp * 2 + (lo ? 1 : 0)) * FFN4_TOK + (wave & 1) * 16) * 32;
            if (kp < 8) ffn_dma16(r_ax, aoff, kp * pstride, dst);
            else ffn_dma16(r_am, aoff, (kp - 8) * pstride, dst);
        }
    };
    issue_chunk(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 1; d < FFN4_D1; ++d) load_w1(d, wq[d]);
#pragma unroll
    for (int c = 1; c < FFN4_LEADC; ++c) issue_chunk(c);
    __builtin_amdgcn_sched_barrier(0);
    ffn_wait_vm<ffn4_ops_after_chunk(0)>();
    __builtin_amdgcn_s_barrier();

    // ------------------------------------------------------------------ phase 1: h^T = W1 . a^T, 4 j tiles x 32 tokens per wave
    f32x16 c1[4], c2[4];                           // [2 slot + jt]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { c1[i][r] = 0.0f; c2[i][r] = 0.0f; }
    const int fsw = (lr >> 2) & 3;
    auto read_a = [&](int ks, half8& ah, half8& al) {
        const int kp = ks >> 1, s = ks & 1;
        const _Float16* base = smem + (kp * 2 * FFN4_TOK + lr) * 32 + (((2 * s + h) ^ fsw) * 8);
        ah = *reinterpret_cast<const half8*>(base);
        al = *reinterpret_cast<const half8*>(base + FFN4_TOK * 32);
    };
    auto mma1 = [&](const half8 (&w)[8], const half8& ah, const half8& al) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const int i = 2 * sl + jt;
                c1[i] = mfma16(w[4 * sl + jt], ah, c1[i]);
                c2[i] = mfma16(w[4 * sl + jt], al, c2[i]);
                c2[i] = mfma16(w[4 * sl + 2 + jt], ah, c2[i]);
            }
    };
    {
        half8 ah0, al0, ah1, al1;
        read_a(0, ah0, al0);
        static_assert(32 % FFN4_D1 == 0 && FFN4_D1 % 2 == 0, "the ring of W1 fragment sets divides the 32 steps, even depth");
        ffn_static_for([&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value, u = ks % FFN4_D1;
            if constexpr (ks % 4 == 3 && (ks + 1) / 4 < 8) {
                ffn_wait_vm<ffn4_ops_after_chunk((ks + 1) / 4)>();
                __builtin_amdgcn_s_barrier();
            }
            if constexpr (ks & 1) {
                if constexpr (ks + 1 < 32) read_a(ks + 1, ah0, al0);
                mma1(wq[u], ah1, al1);
            } else {
                read_a(ks + 1, ah1, al1);
                mma1(wq[u], ah0, al0);
            }
            if constexpr (ks + FFN4_D1 < 32) load_w1(ks + FFN4_D1, wq[u]);
            if constexpr (ks % 4 == 0 && ks / 4 + FFN4_LEADC < 8) issue_chunk(ks / 4 + FFN4_LEADC);
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, 32>{});
    }

    half8 vq[FFN4_D2][4];
#pragma unroll
    for (int d = 0; d < FFN4_D2; ++d) load_w2(d, vq[d]);

    // ------------------------------------------------------------------ LayerNorm + GELU + split, in registers
    // v[2 slot + jt][r] = h[j = 64 (2 wave + slot) + 32 jt + (r & 3) + 8 (r >> 2) + 4 h][tok = lr]
    float v[4][16];
    {
        float s[2] = {0.0f, 0.0f};                 // per slot: the 8-wave kernel's per-wave partial sum
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b = *reinterpret_cast<const float4*>(cst + 64 * (2 * wave + sl) + 32 * jt + 8 * g + 4 * h);
                    const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e, i = 2 * sl + jt;
                        v[i][r] = (c1[i][r] + c2[i][r] * SPLIT_INV) + bb[e];
                        s[sl] += v[i][r];
                    }
                }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            s[sl] += __shfl_xor(s[sl], 32);
            if (h == 0) red[(2 * wave + sl) * FFN4_TOK + lr] = s[sl];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // (also: every wave is done with the operand tile)
        float mean, rstd;
        {
            const float* q = red + lr;
            mean = (((q[0] + q[FFN4_TOK]) + (q[2 * FFN4_TOK] + q[3 * FFN4_TOK])) +
                    ((q[4 * FFN4_TOK] + q[5 * FFN4_TOK]) + (q[6 * FFN4_TOK] + q[7 * FFN4_TOK]))) / 512.0f;
        }
        float qs[2] = {0.0f, 0.0f};
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const int i = 2 * sl + jt; v[i][r] -= mean; qs[sl] += v[i][r] * v[i][r]; }
        float* red2 = red + 8 * FFN4_TOK;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            qs[sl] += __shfl_xor(qs[sl], 32);
            if (h == 0) red2[(2 * wave + sl) * FFN4_TOK + lr] = qs[sl];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            const float* q = red2 + lr;
            const float var = (((q[0] + q[FFN4_TOK]) + (q[2 * FFN4_TOK] + q[3 * FFN4_TOK])) +
                               ((q[4 * FFN4_TOK] + q[5 * FFN4_TOK]) + (q[6 * FFN4_TOK] + q[7 * FFN4_TOK]))) / 512.0f;
            rstd = 1.0f / sqrtf(var + 1e-5f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[i][r] *= rstd;
    }
    // g = GELU(v * gamma + beta), split, out to LDS as the B fragments of phase 2: image [plane][ks 32][lane 64][8]
    float amax = 0.0f;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int slot = 2 * wave + sl, i = 2 * sl + jt;
                const int jb = 64 * slot + 32 * jt + 16 * s + 4 * h;
                const float4 gm0 = *reinterpret_cast<const float4*>(cst + FFN_H + jb), gm1 = *reinterpret_cast<const float4*>(cst + FFN_H + jb + 8);
                const float4 bt0 = *reinterpret_cast<const float4*>(cst + 2 * FFN_H + jb), bt1 = *reinterpret_cast<const float4*>(cst + 2 * FFN_H + jb + 8);
                const float gmm[8] = {gm0.x, gm0.y, gm0.z, gm0.w, gm1.x, gm1.y, gm1.z, gm1.w};
                const float btt[8] = {bt0.x, bt0.y, bt0.z, bt0.w, bt1.x, bt1.y, bt1.z, bt1.w};
                const int ks = 2 * (2 * slot + jt) + s;
                float ge[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y = v[i][8 * s + e] * gmm[e] + btt[e];
                    ge[e] = 0.5f * y * (1.0f + ffn_erf(y * 0.70710678118654752440f));
                }
                uint4 hi, lo4;
                split8_fast(ge, hi, lo4, amax);
                _Float16* dst = smem + (ks * 64 + lane) * 8;
                *reinterpret_cast<uint4*>(dst) = hi;
                *reinterpret_cast<uint4*>(dst + 32 * 64 * 8) = lo4;
            }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // the hidden fragments of all waves are in place

    // ------------------------------------------------------------------ phase 2: y^T = W2 . g^T, wave w = columns [64 w, +64)
    f32x16 d1[2], d2[2];                           // [output slot]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { d1[i][r] = 0.0f; d2[i][r] = 0.0f; }
    auto read_g = [&](int ks, half8& gh, half8& gl) {
        const _Float16* base = smem + (ks * 64 + lane) * 8;
        gh = *reinterpret_cast<const half8*>(base);
        gl = *reinterpret_cast<const half8*>(base + 32 * 64 * 8);
    };
    auto mma2 = [&](const half8 (&w)[4], const half8& gh, const half8& gl) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            d1[sl] = mfma16(w[2 * sl], gh, d1[sl]);
            d2[sl] = mfma16(w[2 * sl], gl, d2[sl]);
            d2[sl] = mfma16(w[2 * sl + 1], gh, d2[sl]);
        }
    };
    {
        half8 gh0, gl0, gh1, gl1;
        read_g(0, gh0, gl0);
        static_assert(32 % FFN4_D2 == 0 && FFN4_D2 % 2 == 0, "the ring of W2 fragment sets divides the 32 steps, even depth");
        for (int ks0 = 0; ks0 < 32; ks0 += FFN4_D2) {
#pragma unroll
            for (int u = 0; u < FFN4_D2; ++u) {
                const int ks = ks0 + u;
                if (u & 1) {
                    if (ks + 1 < 32) read_g(ks + 1, gh0, gl0);
                    mma2(vq[u], gh1, gl1);
                } else {
                    read_g(ks + 1, gh1, gl1);
                    mma2(vq[u], gh0, gl0);
                }
                if (ks + FFN4_D2 < 32) load_w2(ks + FFN4_D2, vq[u]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ------------------------------------------------------------------ epilogue
    constexpr int UNITS = FFN4_TOK * (FFN_D / 8) / 256;      // 4 units of 8 columns per thread
    float4 xa[UNITS], xb[UNITS];
    const bool heads = p.hm != nullptr, with_conf = heads && p.hc != nullptr;
    float hmw[8] = {}, hcw[8] = {};
    if (heads) {
        const float* hw = cst + 3 * FFN_H + FFN_D + (t & 31) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { hmw[e] = hw[e]; hcw[e] = hw[FFN_D + e]; }
    }
    int n_unconf = 0;
#pragma unroll
    for (int it = 0; it < UNITS; ++it) {
        const int u = t + 256 * it, tok = u >> 5, col = (u & 31) * 8;
        const float* xr = p.x + (size_t)(grow0 + min(tok, max(n_valid - 1, 0))) * FFN_D + col;
        xa[it] = *reinterpret_cast<const float4*>(xr); xb[it] = *reinterpret_cast<const float4*>(xr + 4);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // every wave is done reading the hidden fragments
    if (t == 0) *reinterpret_cast<int*>(red) = 0;
    float* const ybuf = reinterpret_cast<float*>(smem_b);
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 o;
            o.x = d1[sl][4 * g] + d2[sl][4 * g] * SPLIT_INV; o.y = d1[sl][4 * g + 1] + d2[sl][4 * g + 1] * SPLIT_INV;
            o.z = d1[sl][4 * g + 2] + d2[sl][4 * g + 2] * SPLIT_INV; o.w = d1[sl][4 * g + 3] + d2[sl][4 * g + 3] * SPLIT_INV;
            *reinterpret_cast<float4*>(ybuf + lr * FFN_Y_LD + 32 * (2 * wave + sl) + 8 * g + 4 * h) = o;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int it = 0; it < UNITS; ++it) {
        const int u = t + 256 * it, tok = u >> 5, col = (u & 31) * 8;
        if (tok >= n_valid) continue;
        const float4 ya = *reinterpret_cast<const float4*>(ybuf + tok * FFN_Y_LD + col);
        const float4 yb = *reinterpret_cast<const float4*>(ybuf + tok * FFN_Y_LD + col + 4);
        const float4 ba = *reinterpret_cast<const float4*>(cst + 3 * FFN_H + col), bb = *reinterpret_cast<const float4*>(cst + 3 * FFN_H + col + 4);
        float o[8] = {(ya.x + ba.x) + xa[it].x, (ya.y + ba.y) + xa[it].y, (ya.z + ba.z) + xa[it].z, (ya.w + ba.w) + xa[it].w,
                      (yb.x + bb.x) + xb[it].x, (yb.y + bb.y) + xb[it].y, (yb.z + bb.z) + xb[it].z, (yb.w + bb.w) + xb[it].w};
        float* xr = p.x + (size_t)(grow0 + tok) * FFN_D + col;
        *reinterpret_cast<float4*>(xr) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(xr + 4) = make_float4(o[4], o[5], o[6], o[7]);
        uint4 hi, lo4;
        split8_fast(o, hi, lo4, amax);
        const size_t po = panel_index(grow0 + tok, col, p.plane_rows);
        *reinterpret_cast<uint4*>(p.xo_hi + po) = hi;
        *reinterpret_cast<uint4*>(p.xo_lo + po) = lo4;
        if (heads) {
            float sm = ((o[0] * hmw[0] + o[1] * hmw[1]) + (o[2] * hmw[2] + o[3] * hmw[3])) +
                       ((o[4] * hmw[4] + o[5] * hmw[5]) + (o[6] * hmw[6] + o[7] * hmw[7]));
            float sc = ((o[0] * hcw[0] + o[1] * hcw[1]) + (o[2] * hcw[2] + o[3] * hcw[3])) +
                       ((o[4] * hcw[4] + o[5] * hcw[5]) + (o[6] * hcw[6] + o[7] * hcw[7]));
            sm = ffn_half_wave_sum(sm);
            if (with_conf) sc = ffn_half_wave_sum(sc);
            if ((t & 31) == 31) {
                p.mat[grow0 + tok] = sm + p.hm_b[0];
                if (with_conf) {
                    const float c = 1.0f / (1.0f + expf(-(sc + p.hc_b[0])));
                    p.conf[grow0 + tok] = c;
                    n_unconf += c < p.conf_thr;
                }
            }
        }
    }
    split_range_check(amax, range_flag);
    if (with_conf && p.unconf) {
        const unsigned long long any = __ballot(n_unconf != 0);
        if (any) {
            int wsum = n_unconf + __shfl_xor(n_unconf, 32);
            if (lane == 31) atomicAdd(reinterpret_cast<int*>(red), wsum);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t == 0) { const int tot = *reinterpret_cast<volatile int*>(red); if (tot) atomicAdd(p.unconf, tot); }
    }
}

}  // namespace sslam
