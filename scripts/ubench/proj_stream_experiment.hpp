// EXPERIMENT RECORD (r03, not part of the product; proj_stream_experiment.patch is the product-side diff it was measured
// with).  Result: bit-identical to the 128 x 128 LDS-ring projection kernels and NOT faster - as one 8-wave workgroup per
// CU QKV 68.8 / cross 44.4 us per 8-pair launch against 60.2 / 41.6; as the 4-wave form below, two workgroups per CU:
// 59.3 / 40.5.  The k-loops alone take 34.7 / 24.3 us; the rest is the 100 MB (QKV) / 67 MB (cross) of split planes the
// epilogue writes: the projections are bound by their output side, not by how the operands reach the matrix pipe
// (delaying half of the first dispatch round to de-phase the bursts only added the delay).
//
// proj_stream.hpp - main loop of the two projections of a LightGlue transformer block (QKV, shared-qk cross; K = 256) in
// the form of the fused FFN's phase 1 (ffn_fused.hpp): the 64-token operand tile (x planes, 64 x 256 hi + lo = 64 KB)
// is fetched ONCE by LDS-DMA and stays, the weights are stored in HBM in FRAGMENT order and stream straight from L2
// into registers (`buffer_load_dwordx4`, a ring of D1 fragment sets per wave, no LDS, no barrier in the k-loop beyond
// one per operand chunk).  The product is the TRANSPOSED one: weight rows are the MFMA's A operand, token rows its B
// operand - accumulator tile (jt, tt) of wave w holds out[token 32 tt + (lane & 31)][column 64 w + 32 jt + row(r, lane)].
// A workgroup is FOUR waves = 64 tokens x 256 output columns with 66 KB of LDS, so two of them share a CU and one's
// epilogue (10 us per 64 x 512 outputs: LDS transposition, split, rotary, V^T scatter) runs under the other's k-loop -
// as ONE 8-wave workgroup per CU the same loop left the matrix pipe idle through every epilogue (cross 44 us per 8-pair
// launch against 24.5 for the loops alone).
//
// r03: the 128 x 128 LDS-ring kernels these replace for batched token sets pull W AND the activations through LDS
// (every fragment a ds_read, 85 B/clk/CU of LDS reads beside the DMA writes) and ran at 26 % of the executed f16
// rate (QKV 60 us, cross 42 us per 8-pair launch); here only the activations go through LDS and every weight byte
// enters the CU once per 64 tokens from the XCD's L2, where all workgroups share the same 0.4 - 0.8 MB.
//
// Arithmetic: every output accumulates its 16 k-steps in ascending k with the three products of a step in the order
// of the ring kernels (hi.hi -> c1; a_hi.w_lo, then a_lo.w_hi -> c2), so the results are theirs bit for bit.
#pragma once
#include <utility>
#include "ffn_fused.hpp"

namespace sslam {

constexpr int PRJ_TOK = 64, PRJ_K = 256;
constexpr int PRJ_STEPS = PRJ_K / 16, PRJ_CHUNKS = PRJ_K / 64;     // 16 k-steps; operand chunks of two 32-deep k-panels
constexpr int PRJ_OPER_BYTES = 8 * 2 * 64 * 32 * 2;                // operand tile [8 k-panels][2 planes][64 tok][32 halves]
#ifndef PRJ_D1
#define PRJ_D1 4            // weight fragment sets in flight per wave
#endif
#ifndef PRJ_LEADC_N
#define PRJ_LEADC_N 2       // operand chunks in flight ahead of the MFMAs that read them
#endif
constexpr int PRJ_LEADC = PRJ_LEADC_N;

constexpr int PRJ_WAVES = 4, PRJ_COLS = 256;                       // waves per workgroup, output columns per workgroup
// Fragment-order layout of one 256-row block of a projection weight W [256][256] (both planes in one buffer): step ks
// (16 k's), wave w (64 rows), then [plane][j tile][lane][8 halves]; lane (h = lane >> 5, lr = lane & 31) holds
// W[64 w + 32 jt + lr][16 ks + 8 h + e].
__host__ __device__ inline size_t prj_frag_index(int plane, int j, int k) {
    const int w = j >> 6, jt = (j >> 5) & 1, lr = j & 31, ks = k >> 4, h = (k >> 3) & 1, e = k & 7;
    return ((((size_t)(ks * PRJ_WAVES + w) * 2 + plane) * 2 + jt) * 64 + (h * 32 + lr)) * 8 + e;
}

// vector-memory operations a wave issues after the last piece of operand chunk c and before the point where chunk c
// must be in place (ffn_ops_after_chunk for this loop: L loads per weight set).  Program order: prologue = weight set 0,
// chunk 0 (P pieces), sets 1 .. D1 - 1, chunks 1 .. LEADC - 1; step s = [wait point] MFMAs, set refill (while
// s + D1 < STEPS), then chunk s / 4 + LEADC when s % 4 == 0.
template <int L, int P>
constexpr int prj_ops_after_chunk(int c) {
    int n = 0; bool seen = false;
    for (int k = 0; k < PRJ_LEADC; ++k) {
        if (seen) n += P;
        if (k == c) seen = true;
        if (k == 0 && seen) n += L * (PRJ_D1 - 1);
    }
    if (c == 0) return n;
    for (int s = 0; s < PRJ_STEPS; ++s) {
        if (s == 4 * c - 1) return n;
        if (s + PRJ_D1 < PRJ_STEPS && seen) n += L;
        if (s % 4 == 0 && s / 4 + PRJ_LEADC < PRJ_CHUNKS) { if (seen) n += P; if (s / 4 + PRJ_LEADC == c) seen = true; }
    }
    return n;
}

// c1 / c2 [j tile][token tile] of this wave for the 64 tokens whose plane rows start at grow0 (rows are clamped to
// grow_cap - 1).  256 threads; `smem`: PRJ_OPER_BYTES of LDS, 16-byte aligned.  `wf`: this workgroup's 256-row weight
// block in fragment order (prj_frag_index), 2 x 256 x 256 halves.  On return every wave has passed its last fragment
// read (the caller's barrier frees the tile).
template <int JT> struct PrjAcc { f32x16 c1[JT][2], c2[JT][2]; };

// (TAG: one specialization per caller - hipcc's host pass drops the definition of a second kernel that calls a device
//  function template specialization another kernel instantiated first)
template <int JT, int TAG>
__device__ __forceinline__ void prj_stream_mainloop(SplitPtr xs, int plane_rows, const _Float16* wf, int grow0, int grow_cap,
                                                    _Float16* smem, PrjAcc<JT>& acc) {
    f32x16 (&c1)[JT][2] = acc.c1;
    f32x16 (&c2)[JT][2] = acc.c2;
    static_assert(JT == 2, "64 output columns per wave");
    static_assert(PRJ_STEPS % PRJ_D1 == 0 && PRJ_D1 % 2 == 0, "the ring of weight sets divides the steps, even depth");
    constexpr int L = 2 * JT;                                  // fragment loads per weight set: [hi jt.., lo jt..]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int h = lane >> 5, lr = lane & 31;
    const int lane16 = lane * 16;
    const auto r_w = ffn_rsrc(wf, 2u * PRJ_COLS * PRJ_K * 2);
    auto load_w = [&](int ks, half8 (&dst)[L]) {
        const int base = (ks * PRJ_WAVES + wave) * (L * 1024);
#pragma unroll
        for (int f = 0; f < L; ++f) dst[f] = ffn_ldfrag(r_w, lane16, base + f * 1024);
    };
    half8 wq[PRJ_D1][L];
    load_w(0, wq[0]);
    // operand tile by LDS-DMA: wave w brings rows 16 w .. + 15 of both planes of every k-panel (4 pieces per chunk)
    constexpr int P = 4;
    const int prow = lane >> 2, pc = lane & 3;
    const int psw = (pc ^ ((prow >> 2) & 3)) * 8;
    const int aoff = (min(grow0 + wave * 16 + prow, grow_cap - 1) * PANEL_K + psw) * 2;            // bytes
    const auto r_ah = ffn_rsrc(xs.hi, (unsigned)plane_rows * PRJ_K * 2), r_al = ffn_rsrc(xs.lo, (unsigned)plane_rows * PRJ_K * 2);
    const int pstride = plane_rows * (PANEL_K * 2);            // bytes per k-panel of a plane
    auto issue_chunk = [&](int c) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kp = 2 * c + j;
            ffn_dma16(r_ah, aoff, kp * pstride, smem + ((kp * 2 + 0) * 64 + wave * 16) * 32);
            ffn_dma16(r_al, aoff, kp * pstride, smem + ((kp * 2 + 1) * 64 + wave * 16) * 32);
        }
    };
    issue_chunk(0);
    __builtin_amdgcn_sched_barrier(0);                         // (the queue order is the point: set 0, chunk 0, then the rest)
#pragma unroll
    for (int d = 1; d < PRJ_D1; ++d) load_w(d, wq[d]);
#pragma unroll
    for (int c = 1; c < PRJ_LEADC; ++c) issue_chunk(c);
    __builtin_amdgcn_sched_barrier(0);
    ffn_wait_vm<prj_ops_after_chunk<L, P>(0)>();
    __builtin_amdgcn_s_barrier();

#pragma unroll
    for (int i = 0; i < JT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { c1[i][j][r] = 0.0f; c2[i][j][r] = 0.0f; }
    const int fsw = (lr >> 2) & 3;
    auto read_a = [&](int ks, half8 (&ah)[2], half8 (&al)[2]) {
        const int kp = ks >> 1, s = ks & 1;
        const _Float16* base = smem + (kp * 2 * 64 + lr) * 32 + (((2 * s + h) ^ fsw) * 8);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            ah[tt] = *reinterpret_cast<const half8*>(base + tt * 32 * 32);
            al[tt] = *reinterpret_cast<const half8*>(base + 64 * 32 + tt * 32 * 32);
        }
    };
    auto mma = [&](const half8 (&w)[L], const half8 (&ah)[2], const half8 (&al)[2]) {
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                c1[jt][tt] = mfma16(w[jt], ah[tt], c1[jt][tt]);
                c2[jt][tt] = mfma16(w[JT + jt], ah[tt], c2[jt][tt]);        // a_hi . w_lo
                c2[jt][tt] = mfma16(w[jt], al[tt], c2[jt][tt]);             // a_lo . w_hi
            }
    };
    half8 ah0[2], al0[2], ah1[2], al1[2];
    read_a(0, ah0, al0);
    ffn_static_for([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value, u = ks % PRJ_D1;
        if constexpr (ks % 4 == 3 && (ks + 1) / 4 < PRJ_CHUNKS) {
            ffn_wait_vm<prj_ops_after_chunk<L, P>((ks + 1) / 4)>();
            __builtin_amdgcn_s_barrier();
        }
        if constexpr (ks & 1) {
            if constexpr (ks + 1 < PRJ_STEPS) read_a(ks + 1, ah0, al0);
            mma(wq[u], ah1, al1);
        } else {
            read_a(ks + 1, ah1, al1);
            mma(wq[u], ah0, al0);
        }
        if constexpr (ks + PRJ_D1 < PRJ_STEPS) load_w(ks + PRJ_D1, wq[u]);
        if constexpr (ks % 4 == 0 && ks / 4 + PRJ_LEADC < PRJ_CHUNKS) issue_chunk(ks / 4 + PRJ_LEADC);
        __builtin_amdgcn_sched_barrier(0);
    }, std::make_integer_sequence<int, PRJ_STEPS>{});
}

}  // namespace sslam
