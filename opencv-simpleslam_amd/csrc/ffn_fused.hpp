// ffn_fused.hpp - the whole FFN of a LightGlue transformer block as ONE kernel (gfx950, split precision).
//
//     x += W2 . GELU(LayerNorm(W1 . [x | message] + b1)) + b2          (oracle/lightglue_ref.py:73-77 `_ffn`;
//                                                                       upstream TransformerLayer.ffn / CrossBlock.ffn)
//
// r02 ran this as two launches (a 64 x 512 whole-row GEMM with LayerNorm + GELU in its epilogue, then a
// 128 x 128 GEMM with the residual): the 512-wide hidden went out to HBM as fp16 plane pairs (67 MB per
// 8-pair launch) and came back, and both kernels pulled every weight tile through an LDS ring whose
// barriers and restart latency - not the matrix pipe - set their time.  Here a workgroup (8 waves, one
// per CU) owns 64 tokens end to end:
//
//   * ACTIVATIONS go through LDS, WEIGHTS never do.  The 64-token operand tile [x | message] (64 x 512,
//     hi + lo planes = 128 KB) is fetched ONCE by LDS-DMA and stays; W1 and W2 are stored in HBM in
//     FRAGMENT order - the 1 KiB a wave needs for one MFMA operand (64 lanes x 8 halves) is contiguous -
//     and stream straight into registers with `buffer_load_dwordx4` (scalar base + scalar step offset +
//     one loop-invariant lane offset: no address arithmetic), each wave its own rows, several steps
//     ahead.  Every weight byte enters the CU exactly once, whole 128-byte lines, and the k-loops have
//     no barrier and no LDS ring.
//   phase 1   h^T[j][tok] = W1[j][:] . a[tok][:]   the TRANSPOSED product: W1 rows are the MFMA's A operand,
//             token rows (LDS) its B operand, so the accumulator has the token on the LANE and the hidden
//             index j in the REGISTERS.  Wave w owns j in [64 w, +64) x all 64 tokens: 2 x 2 tiles of
//             32 x 32, two fp32 accumulators each (hi.hi and the cross terms).
//   LN/GELU   LayerNorm statistics of a token: sums over the registers of its lane, + the other lane half,
//             + the 8 waves through 4 KB of LDS.  GELU and the hi/lo split run in registers.
//   phase 2   y^T[n][tok] = W2[n][:] . g[tok][:] sums over j = the ROW index of the phase-1 tile: registers
//             8 s .. 8 s + 7 of a tile ARE a B fragment of this product (the k order inside a 16-step is the
//             accumulator's row order; W2's columns are stored in that order).  Each wave writes its 16
//             fragments per plane into the LDS region the operand tile occupied (fragment-major, 128 KB),
//             one barrier, and then owns output columns [32 w, +32) over the FULL k = 512: no K split, no
//             partial sums, W2 again straight into registers.
//   epilogue  y staged through LDS; every thread finishes 8-column units of whole rows: + b2, + residual,
//             the fp32 state and its split planes as 16-byte pieces.
// Barriers per tile: 6 (v1 with LDS rings for the weights: 47).  The kernel is bound by the L2 -> CU path
// (1.5 MB of weights + 128 KB of activations per 64 tokens at ~36-46 B/clk/CU), not by the matrix pipe.
#pragma once
#include <utility>
#include "gemm_f16x3.hpp"

namespace sslam {

constexpr int FFN_TOK = 64;                    // tokens per workgroup
constexpr int FFN_D = 256, FFN_H = 512;        // model width, hidden width
constexpr int FFN_OPER_BYTES = 131072;         // operand tile [16 k-panels][2 planes][64 tok][32] = hidden fragments [2][32][2][64][8]
constexpr int FFN_RED_OFF = FFN_OPER_BYTES;    // bytes: LayerNorm partial sums [2 passes][8 waves][64 tok] fp32
constexpr int FFN_CONST_OFF = FFN_RED_OFF + 2 * 8 * FFN_TOK * 4;      // bytes: b1 | ln_w | ln_b (512 each) | b2 (256), fp32
constexpr int FFN_LDS_BYTES = FFN_CONST_OFF + (3 * FFN_H + 3 * FFN_D) * 4;  // 144 384 (+ the two token-head weight rows, 256 each)
constexpr int FFN_Y_LD = 260;                  // fp32 row stride of the staged output tile [64][256]
static_assert(FFN_TOK * FFN_Y_LD * 4 <= FFN_OPER_BYTES, "the output tile is staged where the hidden fragments were");

// accumulator row -> position inside a 16-deep k step: element e (0..7) of lane half h of the B fragment
// built from accumulator registers 8 s .. 8 s + 7 is row 16 s + (e & 3) + 8 (e >> 2) + 4 h of the tile;
// the MFMA calls that slot k = 8 h + e.  W2's column for slot k of a 16-group:
__host__ __device__ inline int ffn_kperm(int k) { return (k & 3) + 8 * ((k >> 2) & 1) + 4 * (k >> 3); }

// Fragment-order layout of W1 [512][512] (both planes in one buffer): step ks (16 k's), wave w (64 rows),
// then [plane][j tile][lane][8 halves]; lane (h = lane >> 5, lr = lane & 31) holds W1[64 w + 32 jt + lr][16 ks + 8 h + e].
__host__ __device__ inline size_t ffn_w1_frag_index(int plane, int j, int k) {
    const int w = j >> 6, jt = (j >> 5) & 1, lr = j & 31, ks = k >> 4, h = (k >> 3) & 1, e = k & 7;
    return ((((size_t)(ks * 8 + w) * 2 + plane) * 2 + jt) * 64 + (h * 32 + lr)) * 8 + e;
}
// Fragment-order layout of W2 [256][512]: step ks, wave w (32 rows), [plane][lane][8 halves]; lane holds
// W2[32 w + lr][16 ks + ffn_kperm(8 h + e)].
__host__ __device__ inline size_t ffn_w2_frag_index(int plane, int n, int j) {
    const int w = n >> 5, lr = n & 31, ks = j >> 4, r = j & 15;
    const int k = (r & 3) + 4 * ((r >> 3) & 1) + 8 * ((r >> 2) & 1);        // inverse of ffn_kperm
    return (((size_t)(ks * 8 + w) * 2 + plane) * 64 + ((k >> 3) * 32 + lr)) * 8 + (k & 7);
}

struct FfnFusedArgs {
    SplitPtr xs, msgs;          // token state / attention context planes, k-panel layout (PANEL_K = 32) over plane_rows rows
    int plane_rows;
    const _Float16* w1f;        // W1, both planes, fragment order (ffn_w1_frag_index): 1 MB
    const float* b1; const float* ln_w; const float* ln_b;
    const _Float16* w2f;        // W2, both planes, fragment order (ffn_w2_frag_index): 512 KB
    const float* b2;
    float* x;                   // fp32 token state [plane_rows][256]: residual in, new state out
    _Float16* xo_hi; _Float16* xo_lo;   // split planes of the new state (k-panel layout; may alias xs)
    unsigned long long* stamps;         // diagnostic builds only (FFN_STAMP): [workgroup][8] s_memtime values; else unused
    // token heads on the new state (LightGlue's token-confidence and matchability Linear(256, 1), evaluated after a
    // layer's cross block; lightglue.py check_if_stop / get_pruning_mask): hm != nullptr turns them on.
    //   mat[row] = hm . x + hm_b ; hc != nullptr: conf[row] = sigmoid(hc . x + hc_b), *unconf += #(conf < conf_thr)
    const float* hm; const float* hm_b; const float* hc; const float* hc_b;
    float conf_thr; float* conf; float* mat; int* unconf;
    // FOLD form (r05, one pair): the attention ran over `ks` key ranges and left unnormalised partials (o, m, l) per range
    // (lg_attn_merge_h_kernel's inputs); the tile merges its own 32 tokens in the prologue instead of reading `msgs`.
    // partial of (range z, head hd, token tok): index z * part_zs + part_base + hd * part_kc + tok, o_part 64 floats each.
    const float* o_part; const float* m_part; const float* l_part;
    int ks; int part_kc; long part_zs; long part_base;
};

#ifdef FFN_STAMP
// stamps are kept in registers and stored once at the end of the tile: a store in the middle of the k-loops would be one
// more vector-memory operation in the wave's in-order queue and throw the counted waits off
#define FFN_STAMP_AT(i_) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ffn_st_[(i_)] = t_; } while (0)
#else
#define FFN_STAMP_AT(i_)
#endif
#ifndef FFN_ABL
#define FFN_ABL 0       // ubench ablations: 1 no MFMA, 2 no weight loads after the first steps, 4 no LN/GELU arithmetic, 8 no stores
#endif
#ifndef FFN_D1
#define FFN_D1 4        // W1 fragment sets in flight per wave (steps of 4 KB)
#endif
#ifndef FFN_D2
#define FFN_D2 8        // W2 fragment sets in flight per wave (steps of 2 KB)
#endif
// Ring depths of the 32-token tile (TT = 1: one or two pairs, pruned sets).  r05 measured twice the depth there (8 / 16 sets:
// the tile's accumulators and LayerNorm values need 96 registers less, so it fits - 255 VGPRs, no spill - and the results are
// bit-identical): 25.6 against 24.3 us per launch for one pair, 32.8 against 31.4 for two.  The 32-token tile does not wait
// for its weights; what it cannot hide is the prologue / LayerNorm-GELU / epilogue chain that runs without an MFMA beside it.
#ifndef FFN_D1_SMALL
#define FFN_D1_SMALL FFN_D1
#endif
#ifndef FFN_D2_SMALL
#define FFN_D2_SMALL FFN_D2
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
// one fragment: 64 lanes x 16 B straight into registers
template <typename R>
__device__ __forceinline__ half8 ffn_ldfrag(R rsrc, int voff_bytes, int soff_bytes) {
    return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_bytes, soff_bytes, 0));
}

// sum over each 32-lane half of the wave, valid in lanes 31 and 63 (row_shr 1 / 2 / 3, row_shr 4 and 8 on the upper
// banks, then lane 15 of rows 0 / 2 added into rows 1 / 3; lanes without a source add 0)
__device__ __forceinline__ float ffn_half_wave_sum(float x) {
    auto dpp = [](float v, auto ctrl, auto row_mask, auto bank_mask) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value,
                                                                     decltype(row_mask)::value, decltype(bank_mask)::value, false));
    };
    using std::integral_constant;
    float v = x + dpp(x, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{});
    v += dpp(x, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{});
    v += dpp(x, integral_constant<int, 0x113>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{});
    v += dpp(v, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xe>{});
    v += dpp(v, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xc>{});
    v += dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{}, integral_constant<int, 0xf>{});
    return v;
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

// ---- operand-tile streaming -----------------------------------------------------------------------------------
// r03b: the operand tile arrives in CHUNKS of two k-panels (= four 16-deep steps of phase 1), FFN_LEADC chunks ahead
// of the MFMAs that read them, instead of all 128 KB up front.  Vector-memory operations of a wave return in order, so
// with everything issued in the prologue the first W1 refill a wave had to wait for (step 4) also waited for its last
// operand piece: the whole tile's fetch (13 k of a tile's 77 k cycles, an HBM burst every workgroup issues at the same
// moment) sat in front of the MFMAs.  Now a wave waits for exactly its own pieces of the next chunk with a counted
// s_waitcnt - ffn_ops_after_chunk() replays the wave's issue order at compile time - and one barrier per chunk makes
// the other waves' pieces visible.
#ifndef FFN_LEADC_N
#define FFN_LEADC_N 2
#endif
constexpr int FFN_LEADC = FFN_LEADC_N;
// vector-memory operations a wave issues after the last piece of chunk c and before the point where chunk c must be
// in place (before the loop for chunk 0, else the start of step 4 c - 1: the fragment read runs one step ahead).
// Program order: prologue = W1 set 0 (4 loads), chunk 0 (two pieces), W1 sets 1 .. FFN_D1 - 1, chunks 1 .. LEADC - 1 -
// what step 0 needs first in the queue; step s = [wait point] MFMAs, W1 refill (4 loads, while s + FFN_D1 < 32), then
// chunk s / 4 + LEADC when s % 4 == 0.
constexpr int ffn_ops_after_chunk(int c, int d1, int nch = 8) {     // nch: chunks that arrive by LDS-DMA (FOLD: the x half only)
    int n = 0; bool seen = false;
    for (int k = 0; k < FFN_LEADC; ++k) {
        if (seen) n += 2;
        if (k == c) seen = true;
        if (k == 0 && seen) n += 4 * (d1 - 1);
    }
    if (c == 0) return n;
    for (int s = 0; s < 32; ++s) {
        if (s == 4 * c - 1) return n;
        if (s + d1 < 32 && seen) n += 4;
        if (s % 4 == 0 && s / 4 + FFN_LEADC < nch) { if (seen) n += 2; if (s / 4 + FFN_LEADC == c) seen = true; }
    }
    return n;
}
template <int N> __device__ __forceinline__ void ffn_wait_vm() {
#if FFN_ABL & 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (ablation builds skip vector-memory operations the count assumes)
#else
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
#endif
}
template <class F, int... I> __device__ __forceinline__ void ffn_static_for(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

// One tile of 32 TT tokens (TT = 2: the 64-token tile described above; TT = 1: 32 tokens, for token sets too small
// to give every CU a 64-token tile - one pair, pruned sets: half the MFMA / LayerNorm / GELU work per workgroup and
// twice the workgroups.  The LDS images keep their 64-token strides, the second token tile simply does not exist; a
// token's arithmetic is the same in both forms, bit for bit).  grow0: plane row of token 0; grow_cap: one past the
// last plane row that may be read (rows are clamped to it); n_valid: tokens of the tile that exist (stores are masked
// to them).  512 threads; `smem` = FFN_LDS_BYTES of dynamic LDS, 16-byte aligned.
// FOLD (TT = 1 only): the message half of the operand tile does not come from `msgs` - the tile merges the key-range
// partials of its 32 tokens itself (lg_attn_merge_h_kernel's arithmetic, expression for expression) and writes the planes
// straight into the LDS image: one launch and one HBM round trip of the context planes less per block of a single pair.
template <int TT = 2, bool FOLD = false, int D1 = (TT == 1 ? FFN_D1_SMALL : FFN_D1), int D2 = (TT == 1 ? FFN_D2_SMALL : FFN_D2)>
__device__ __forceinline__ void ffn_fused_tile(const FfnFusedArgs& p, int grow0, int grow_cap, int n_valid,
                                               int* range_flag, _Float16* smem) {
    static_assert(TT == 1 || TT == 2, "one or two 32-token tiles per workgroup");
    static_assert(!FOLD || TT == 1, "the merge prologue is laid out for 32 tokens x 4 heads x 4 quarter-heads = 512 threads");
    constexpr int NCH = FOLD ? 4 : 8;              // operand chunks that arrive by LDS-DMA
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int h = lane >> 5, lr = lane & 31;
    char* const smem_b = reinterpret_cast<char*>(smem);
    float* const red = reinterpret_cast<float*>(smem_b + FFN_RED_OFF);
    const int lane16 = lane * 16;
    float* const cst = reinterpret_cast<float*>(smem_b + FFN_CONST_OFF);
#ifdef FFN_STAMP
    unsigned long long ffn_st_[8] = {};
#endif
    FFN_STAMP_AT(0);
    {   // constants of the two epilogues into LDS (a global load at their point of use is an exposed L2 round trip
        // in front of arithmetic that all eight waves wait for)
        const float b1v = p.b1[t], lwv = p.ln_w[t], lbv = p.ln_b[t];
        const float b2v = p.b2[t & (FFN_D - 1)];
        cst[t] = b1v; cst[FFN_H + t] = lwv; cst[2 * FFN_H + t] = lbv;
        if (t < FFN_D) cst[3 * FFN_H + t] = b2v;
        if (p.hm != nullptr) {                     // token heads: the two weight rows, read back 8 columns per lane in the epilogue
            const bool second = t >= FFN_D;
            const float* src = second ? p.hc : p.hm;
            cst[3 * FFN_H + FFN_D + t] = src != nullptr ? src[t & (FFN_D - 1)] : 0.0f;
        }
    }

    const auto r_w1 = ffn_rsrc(p.w1f, 2 * FFN_H * FFN_H * 2), r_w2 = ffn_rsrc(p.w2f, 2 * FFN_D * FFN_H * 2);
    // W1 fragment set of step ks: [hi jt0, hi jt1, lo jt0, lo jt1], 1 KiB each, 4 KiB per (step, wave)
    auto load_w1 = [&](int ks, half8 (&dst)[4]) {
#if FFN_ABL & 2
        if (ks >= D1) return;
#endif
        const int base = (ks * 8 + wave) * 4096;
#pragma unroll
        for (int f = 0; f < 4; ++f) dst[f] = ffn_ldfrag(r_w1, lane16, base + f * 1024);
    };
    auto load_w2 = [&](int ks, half8 (&dst)[2]) {
#if FFN_ABL & 2
        if (ks >= D2) return;
#endif
        const int base = (ks * 8 + wave) * 2048;
        dst[0] = ffn_ldfrag(r_w2, lane16, base);
        dst[1] = ffn_ldfrag(r_w2, lane16, base + 1024);
    };

    // ------------------------------------------------------------------ prologue
    // the first W1 fragment sets (plain loads, to registers), then the operand tile by LDS-DMA: wave w
    // brings rows 16 (w & 3) .. + 15 of plane (w >> 2) of every k-panel (16 pieces of 16 rows x 64 B)
    // FOLD: thread = (token t >> 4, head (t >> 2) & 3, quarter q = t & 3 of the head's 64 columns); its partials go into
    // the queue FIRST (the oldest operations: every later counted wait covers them)
    float4 po[4][4]; float pm[4], pl[4];
    const int ftok = t >> 4, fhead = (t >> 2) & 3, fq = t & 3;
    if constexpr (FOLD) {
        const size_t rid = (size_t)p.part_base + (size_t)fhead * p.part_kc + min(ftok, max(n_valid - 1, 0));
#pragma unroll
        for (int z = 0; z < 4; ++z) {
            pm[z] = -INFINITY; pl[z] = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) po[z][c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (z < p.ks) {
                const size_t pb = (size_t)z * p.part_zs + rid;
                pm[z] = p.m_part[pb]; pl[z] = p.l_part[pb];
#pragma unroll
                for (int c = 0; c < 4; ++c) po[z][c] = *reinterpret_cast<const float4*>(p.o_part + pb * 64 + fq * 16 + 4 * c);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    half8 wq[D1][4];
    load_w1(0, wq[0]);
    const int prow = lane >> 2, pc = lane & 3;
    const int psw = (pc ^ ((prow >> 2) & 3)) * 8;
    const int aoff = (min(grow0 + (wave & 3) * 16 + prow, grow_cap - 1) * PANEL_K + psw) * 2;      // bytes
    const bool lo = wave >= 4;
    const unsigned a_bytes = (unsigned)p.plane_rows * FFN_D * 2;
    const auto r_ax = ffn_rsrc(lo ? p.xs.lo : p.xs.hi, a_bytes), r_am = ffn_rsrc(lo ? p.msgs.lo : p.msgs.hi, a_bytes);
    const int pstride = p.plane_rows * (PANEL_K * 2);               // bytes per k-panel of a plane
    const bool dma_wave = TT == 2 || (wave & 3) < 2;                // (TT = 1: rows 32 .. 63 of the image are never read)
    // chunk c = k-panels 2 c, 2 c + 1 (panels 0 - 7: x, 8 - 15: message): this wave's two pieces of 16 rows x 64 B each
    auto issue_chunk = [&](int c) {
        if (!dma_wave || c >= NCH) return;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kp = 2 * c + j;
            _Float16* dst = smem + ((kp * 2 + (lo ? 1 : 0)) * 64 + (wave & 3) * 16) * 32;
            if (kp < 8) ffn_dma16(r_ax, aoff, kp * pstride, dst);
            else ffn_dma16(r_am, aoff, (kp - 8) * pstride, dst);
        }
    };
    issue_chunk(0);
    __builtin_amdgcn_sched_barrier(0);                 // (the queue order is the point: set 0, chunk 0, then the rest)
#pragma unroll
    for (int d = 1; d < D1; ++d) load_w1(d, wq[d]);
#pragma unroll
    for (int c = 1; c < FFN_LEADC; ++c) issue_chunk(c);
    __builtin_amdgcn_sched_barrier(0);
    ffn_wait_vm<ffn_ops_after_chunk(0, D1, NCH)>();        // own pieces of chunk 0 (and W1 set 0, older) have landed
    float amax = 0.0f;
    if constexpr (FOLD) {
        // lg_attn_merge_h_kernel: M = max m_z ; acc = sum o_z 2^(m_z - M) ; L = sum l_z 2^(m_z - M) ; context = acc / L
        float M = -INFINITY;
#pragma unroll
        for (int z = 0; z < 4; ++z) M = fmaxf(M, pm[z]);                // (ranges >= ks hold -inf)
        float acc[16]; float L = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int z = 0; z < 4; ++z)
            if (z < p.ks) {
                const float wz = (pm[z] == -INFINITY) ? 0.0f : exp2f(pm[z] - M);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[4 * c] += po[z][c].x * wz; acc[4 * c + 1] += po[z][c].y * wz;
                    acc[4 * c + 2] += po[z][c].z * wz; acc[4 * c + 3] += po[z][c].w * wz;
                }
                L += pl[z] * wz;
            }
        const float inv = 1.0f / L;
        const bool fvalid = ftok < n_valid;            // (rows past the image's tokens: no partials exist - exact zeros)
        float v0[8], v1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { v0[e] = fvalid ? acc[e] * inv : 0.0f; v1[e] = fvalid ? acc[8 + e] * inv : 0.0f; }
        uint4 h0, l0, h1, l1;
        split8_fast(v0, h0, l0, amax);
        split8_fast(v1, h1, l1, amax);
        // columns 64 head + 16 q .. + 15 of the message = k-panel 8 + 2 head + (q >> 1), logical chunks 2 (q & 1), + 1
        const int kp = 8 + 2 * fhead + (fq >> 1), lc = 2 * (fq & 1), sw = (ftok >> 2) & 3;
        _Float16* dh = smem + ((kp * 2) * 64 + ftok) * 32;
        *reinterpret_cast<uint4*>(dh + ((lc ^ sw) * 8)) = h0;
        *reinterpret_cast<uint4*>(dh + (((lc + 1) ^ sw) * 8)) = h1;
        *reinterpret_cast<uint4*>(dh + 64 * 32 + ((lc ^ sw) * 8)) = l0;
        *reinterpret_cast<uint4*>(dh + 64 * 32 + (((lc + 1) ^ sw) * 8)) = l1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    FFN_STAMP_AT(1);

    // ------------------------------------------------------------------ phase 1: h^T = W1 . a^T
    f32x16 c1[2][TT], c2[2][TT];                   // [j tile][token tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { c1[i][j][r] = 0.0f; c2[i][j][r] = 0.0f; }
    const int fsw = (lr >> 2) & 3;
    // B fragments of step ks: token tile tt, planes hi / lo; image [k-panel][plane][tok][32 halves], chunk swizzled
    auto read_a = [&](int ks, half8 (&ah)[TT], half8 (&al)[TT]) {
        const int kp = ks >> 1, s = ks & 1;
        const _Float16* base = smem + (kp * 2 * 64 + lr) * 32 + (((2 * s + h) ^ fsw) * 8);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            ah[tt] = *reinterpret_cast<const half8*>(base + tt * 32 * 32);
            al[tt] = *reinterpret_cast<const half8*>(base + 64 * 32 + tt * 32 * 32);
        }
    };
    auto mma1 = [&](const half8 (&w)[4], const half8 (&ah)[TT], const half8 (&al)[TT]) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
#if FFN_ABL & 1
                c1[jt][tt][0] += (float)w[jt][0] + (float)ah[tt][1];
                c2[jt][tt][0] += (float)w[2 + jt][0] + (float)al[tt][1];
#else
                c1[jt][tt] = mfma16(w[jt], ah[tt], c1[jt][tt]);
                c2[jt][tt] = mfma16(w[jt], al[tt], c2[jt][tt]);
                c2[jt][tt] = mfma16(w[2 + jt], ah[tt], c2[jt][tt]);
#endif
            }
    };
    {
        half8 ah0[TT], al0[TT], ah1[TT], al1[TT];
        read_a(0, ah0, al0);
        static_assert(32 % D1 == 0 && D1 % 2 == 0, "the ring of W1 fragment sets divides the 32 steps, even depth");
        ffn_static_for([&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value, u = ks % D1;
            if constexpr (ks % 4 == 3 && (ks + 1) / 4 < NCH) {
                // the next step's fragments come from chunk (ks + 1) / 4: own pieces landed, then everybody's
                ffn_wait_vm<ffn_ops_after_chunk((ks + 1) / 4, D1, NCH)>();
                __builtin_amdgcn_s_barrier();
            }
            if constexpr (ks & 1) {
                if constexpr (ks + 1 < 32) read_a(ks + 1, ah0, al0);
                mma1(wq[u], ah1, al1);
            } else {
                read_a(ks + 1, ah1, al1);
                mma1(wq[u], ah0, al0);
            }
            if constexpr (ks + D1 < 32) load_w1(ks + D1, wq[u]);
            if constexpr (ks % 4 == 0 && ks / 4 + FFN_LEADC < NCH) issue_chunk(ks / 4 + FFN_LEADC);
            // keep the refill HERE, D1 steps ahead of its use: left alone the scheduler sinks it to just
            // in front of the consuming MFMAs (fewer live registers, one exposed L2 round trip per step)
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, 32>{});
    }
    FFN_STAMP_AT(2);

    // W2 starts streaming now: its first fragment sets land while the LayerNorm / GELU arithmetic runs
    half8 vq[D2][2];
#pragma unroll
    for (int d = 0; d < D2; ++d) load_w2(d, vq[d]);

    // ------------------------------------------------------------------ LayerNorm + GELU + split, in registers
    // v[jt][tt][r] = h[j = 64 w + 32 jt + (r & 3) + 8 (r >> 2) + 4 h][tok = 32 tt + lr]
    float v[2][TT][16];
    {
        float s[TT] = {};
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = *reinterpret_cast<const float4*>(cst + 64 * wave + 32 * jt + 8 * g + 4 * h);
                const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        v[jt][tt][r] = (c1[jt][tt][r] + c2[jt][tt][r] * SPLIT_INV) + bb[e];
                        s[tt] += v[jt][tt][r];
                    }
            }
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            s[tt] += __shfl_xor(s[tt], 32);
            if (h == 0) red[wave * FFN_TOK + 32 * tt + lr] = s[tt];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // (also: every wave is done with the operand tile)
        float mean[TT], rstd[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const float* q = red + 32 * tt + lr;
            mean[tt] = (((q[0] + q[FFN_TOK]) + (q[2 * FFN_TOK] + q[3 * FFN_TOK])) +
                        ((q[4 * FFN_TOK] + q[5 * FFN_TOK]) + (q[6 * FFN_TOK] + q[7 * FFN_TOK]))) / 512.0f;
        }
        float qs[TT] = {};
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { v[jt][tt][r] -= mean[tt]; qs[tt] += v[jt][tt][r] * v[jt][tt][r]; }
        float* red2 = red + 8 * FFN_TOK;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            qs[tt] += __shfl_xor(qs[tt], 32);
            if (h == 0) red2[wave * FFN_TOK + 32 * tt + lr] = qs[tt];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const float* q = red2 + 32 * tt + lr;
            const float var = (((q[0] + q[FFN_TOK]) + (q[2 * FFN_TOK] + q[3 * FFN_TOK])) +
                               ((q[4 * FFN_TOK] + q[5 * FFN_TOK]) + (q[6 * FFN_TOK] + q[7 * FFN_TOK]))) / 512.0f;
            rstd[tt] = 1.0f / sqrtf(var + 1e-5f);
        }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) v[jt][tt][r] *= rstd[tt];
    }
    // g = GELU(v * gamma + beta), split, and out to LDS as the B fragments of phase 2: fragment (plane, ks, tt)
    // = registers 8 s .. 8 s + 7 of tile (jt, tt), ks = 2 (2 w + jt) + s; image [plane][ks 32][tt 2][lane 64][8]
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int jb = 64 * wave + 32 * jt + 16 * s + 4 * h;
            const float4 gm0 = *reinterpret_cast<const float4*>(cst + FFN_H + jb), gm1 = *reinterpret_cast<const float4*>(cst + FFN_H + jb + 8);
            const float4 bt0 = *reinterpret_cast<const float4*>(cst + 2 * FFN_H + jb), bt1 = *reinterpret_cast<const float4*>(cst + 2 * FFN_H + jb + 8);
            const float gmm[8] = {gm0.x, gm0.y, gm0.z, gm0.w, gm1.x, gm1.y, gm1.z, gm1.w};
            const float btt[8] = {bt0.x, bt0.y, bt0.z, bt0.w, bt1.x, bt1.y, bt1.z, bt1.w};
            const int ks = 2 * (2 * wave + jt) + s;
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                float ge[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y = v[jt][tt][8 * s + e] * gmm[e] + btt[e];
#if FFN_ABL & 4
                    ge[e] = y;
#else
                    ge[e] = 0.5f * y * (1.0f + ffn_erf(y * 0.70710678118654752440f));
#endif
                }
                uint4 hi, lo;
                split8_fast(ge, hi, lo, amax);
                _Float16* dst = smem + ((ks * 2 + tt) * 64 + lane) * 8;
                *reinterpret_cast<uint4*>(dst) = hi;
                *reinterpret_cast<uint4*>(dst + 32 * 2 * 64 * 8) = lo;
            }
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // the hidden fragments of all waves are in place
    FFN_STAMP_AT(3);

    // ------------------------------------------------------------------ phase 2: y^T = W2 . g^T, wave w = columns [32 w, +32)
    f32x16 d1[TT], d2[TT];                         // [token tile]
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { d1[i][r] = 0.0f; d2[i][r] = 0.0f; }
    auto read_g = [&](int ks, half8 (&gh)[TT], half8 (&gl)[TT]) {
        const _Float16* base = smem + (ks * 2 * 64 + lane) * 8;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            gh[tt] = *reinterpret_cast<const half8*>(base + tt * 64 * 8);
            gl[tt] = *reinterpret_cast<const half8*>(base + 32 * 2 * 64 * 8 + tt * 64 * 8);
        }
    };
    auto mma2 = [&](const half8 (&w)[2], const half8 (&gh)[TT], const half8 (&gl)[TT]) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
#if FFN_ABL & 1
            d1[tt][0] += (float)w[0][0] + (float)gh[tt][1];
            d2[tt][0] += (float)w[1][0] + (float)gl[tt][1];
#else
            d1[tt] = mfma16(w[0], gh[tt], d1[tt]);
            d2[tt] = mfma16(w[0], gl[tt], d2[tt]);
            d2[tt] = mfma16(w[1], gh[tt], d2[tt]);
#endif
        }
    };
    {
        half8 gh0[TT], gl0[TT], gh1[TT], gl1[TT];
        read_g(0, gh0, gl0);
        static_assert(32 % D2 == 0 && D2 % 2 == 0, "the ring of W2 fragment sets divides the 32 steps, even depth");
        for (int ks0 = 0; ks0 < 32; ks0 += D2) {
#pragma unroll
            for (int u = 0; u < D2; ++u) {
                const int ks = ks0 + u;
                if (u & 1) {
                    if (ks + 1 < 32) read_g(ks + 1, gh0, gl0);
                    mma2(vq[u], gh1, gl1);
                } else {
                    read_g(ks + 1, gh1, gl1);
                    mma2(vq[u], gh0, gl0);
                }
                if (ks + D2 < 32) load_w2(ks + D2, vq[u]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    FFN_STAMP_AT(4);

    // ------------------------------------------------------------------ epilogue
    // residual of this thread's units first (its latency hides under the staging)
    constexpr int UNITS = 32 * TT * (FFN_D / 8) / 512;       // 4 (2) units of 8 columns per thread
    float4 xa[UNITS], xb[UNITS];
    // (token heads: this thread's 8 columns of the two weight rows; a unit's 32 threads are one half wave = one token)
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
        const int u = t + 512 * it, tok = u >> 5, col = (u & 31) * 8;
        const float* xr = p.x + (size_t)(grow0 + min(tok, max(n_valid - 1, 0))) * FFN_D + col;
        xa[it] = *reinterpret_cast<const float4*>(xr); xb[it] = *reinterpret_cast<const float4*>(xr + 4);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // every wave is done reading the hidden fragments
    if (t == 0) *reinterpret_cast<int*>(red) = 0;  // (the workgroup's count of unconfident tokens; the LayerNorm sums are long read)
    float* const ybuf = reinterpret_cast<float*>(smem_b);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 o;
            o.x = d1[tt][4 * g] + d2[tt][4 * g] * SPLIT_INV; o.y = d1[tt][4 * g + 1] + d2[tt][4 * g + 1] * SPLIT_INV;
            o.z = d1[tt][4 * g + 2] + d2[tt][4 * g + 2] * SPLIT_INV; o.w = d1[tt][4 * g + 3] + d2[tt][4 * g + 3] * SPLIT_INV;
            // tile rows 8 g + 4 h .. + 3 = output columns 32 w + 8 g + 4 h .. + 3 of token 32 tt + lr
            *reinterpret_cast<float4*>(ybuf + (32 * tt + lr) * FFN_Y_LD + 32 * wave + 8 * g + 4 * h) = o;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    FFN_STAMP_AT(5);
#pragma unroll
    for (int it = 0; it < UNITS; ++it) {
        const int u = t + 512 * it, tok = u >> 5, col = (u & 31) * 8;
        if (tok >= n_valid) continue;
        const float4 ya = *reinterpret_cast<const float4*>(ybuf + tok * FFN_Y_LD + col);
        const float4 yb = *reinterpret_cast<const float4*>(ybuf + tok * FFN_Y_LD + col + 4);
        const float4 ba = *reinterpret_cast<const float4*>(cst + 3 * FFN_H + col), bb = *reinterpret_cast<const float4*>(cst + 3 * FFN_H + col + 4);
        float o[8] = {(ya.x + ba.x) + xa[it].x, (ya.y + ba.y) + xa[it].y, (ya.z + ba.z) + xa[it].z, (ya.w + ba.w) + xa[it].w,
                      (yb.x + bb.x) + xb[it].x, (yb.y + bb.y) + xb[it].y, (yb.z + bb.z) + xb[it].z, (yb.w + bb.w) + xb[it].w};
        float* xr = p.x + (size_t)(grow0 + tok) * FFN_D + col;
#if !(FFN_ABL & 8)
        *reinterpret_cast<float4*>(xr) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(xr + 4) = make_float4(o[4], o[5], o[6], o[7]);
        uint4 hi, lo;
        split8_fast(o, hi, lo, amax);
        const size_t po = panel_index(grow0 + tok, col, p.plane_rows);
        *reinterpret_cast<uint4*>(p.xo_hi + po) = hi;
        *reinterpret_cast<uint4*>(p.xo_lo + po) = lo;
        if (heads) {
            // 8 columns in the lane ...
            float sm = ((o[0] * hmw[0] + o[1] * hmw[1]) + (o[2] * hmw[2] + o[3] * hmw[3])) +
                       ((o[4] * hmw[4] + o[5] * hmw[5]) + (o[6] * hmw[6] + o[7] * hmw[7]));
            float sc = ((o[0] * hcw[0] + o[1] * hcw[1]) + (o[2] * hcw[2] + o[3] * hcw[3])) +
                       ((o[4] * hcw[4] + o[5] * hcw[5]) + (o[6] * hcw[6] + o[7] * hcw[7]));
            // the 32 lanes of the token's half wave: DPP row reductions (vector ALU; five dependent LDS permutes per value
            // and unit measured 7.6 us per launch of 8 pairs), the total lands in the half's last lane
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
#else
        if (o[0] == 123.456f) *xr = o[1];
#endif
    }
    split_range_check(amax, range_flag);
    if (with_conf && p.unconf) {
        // one atomic per workgroup (same-address atomics retire one per ~11 ns): lanes 31 / 63 of every wave hold counts
        const unsigned long long any = __ballot(n_unconf != 0);
        if (any) {
            int wsum = n_unconf + __shfl_xor(n_unconf, 32);          // (lanes other than 31 / 63 hold 0)
            if (lane == 31) atomicAdd(reinterpret_cast<int*>(red), wsum);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t == 0) { const int tot = *reinterpret_cast<volatile int*>(red); if (tot) atomicAdd(p.unconf, tot); }
    }
    FFN_STAMP_AT(6);
    FFN_STAMP_AT(7);
#ifdef FFN_STAMP
    if (threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) p.stamps[blockIdx.x * 8 + i] = ffn_st_[i];
#endif
}

}  // namespace sslam
