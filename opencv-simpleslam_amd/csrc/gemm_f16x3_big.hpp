// gemm_f16x3_big.hpp - split-precision matrix-core GEMM main loop for BATCHED token sets (gfx950).
//
// The 64-row ring GEMM of gemm_f16x3.hpp is sized for one pair (4096 token rows -> 256 workgroups):
// its 64x128 tile takes in 4 (BM + BN) = 768 bytes of operand planes per k for 2 BM BN = 16 k flop, and
// a CU's LDS-DMA intake - not its matrix pipe - sets the time.  With a batch of pairs there are >= 16 k
// rows, so the tile grows to 128 x 128 (1024 B per k for 32 k flop).
//
// Structure (one workgroup = 4 waves, two workgroups per CU: one's epilogue runs under the other's main loop):
//   * wave (wm, wn) of the 2 x 2 grid owns a 64 x 64 sub-tile: 2 x 2 MFMA tiles of 32 x 32, two fp32
//     accumulators each (hi.hi and the cross terms) = 128 accumulator registers
//   * k-tile = 32 halves (64 B per row per plane, one whole k-panel row: PANEL_K = 32): 2-stage LDS ring
//     of (2 BM + 2 BN) x 64 B = 32 KB per stage, filled by LDS-DMA (global_load_lds_dwordx4: 1 KiB =
//     16 rows x 64 B, contiguous in HBM, per wave instruction; 8 per wave per k-tile)
//   * ONE raw s_barrier per k-tile: [wait own pieces of tile kt] barrier [refill the stage read
//     before the barrier] [ds_read_b128 fragments + 24 MFMA of tile kt]
//   * LDS image un-padded (DMA writes linearly); the 16-byte chunk index is XOR-swizzled with
//     (row >> 2) & 3 on the SOURCE address and on every fragment read: the 16 lanes of a
//     ds_read_b128 group then touch 16 distinct 16-byte slots of the 256-byte bank row.
// (r02 carried a 128 x 256 / 8-wave form with three software-pipelined main loops, a 64 x 512 whole-row form
// with LayerNorm in its epilogue and ablation switches in this file: measured in profiles/r02_gemm_*.txt, kept
// for the ubench as scripts/ubench/gemm_f16x3_big_r02.hpp.  The FFN now runs as ONE kernel, ffn_fused.hpp.)
#pragma once
#include "gemm_f16x3.hpp"

namespace sslam {

constexpr int BBK = 32;                       // halves of k per stage
constexpr int BIG_STAGES = 2;

template <int BM, int BN>
constexpr int big_stage_halves() { return (2 * BM + 2 * BN) * BBK; }

// offset (halves) of row 0 of the 32-deep k-tile that starts at column k (k % 32 == 0) of a plane
__device__ __forceinline__ size_t panel_base(int k, int rows_total) {
    return (size_t)(k / PANEL_K) * rows_total * PANEL_K + (k % PANEL_K);
}

// A planes: k-panel layout over `a_rows` rows; W planes: k-panel layout over `col_cap` rows.
// row0/row_cap, col0/col_cap in plane-row units.  K % 32 == 0, ga.K0 % PANEL_K == 0.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void gemm_mainloop_big(const GemmAH& ga, SplitPtr W, int a_rows, int K, int row0,
                                                  int row_cap, int col0, int col_cap, _Float16* smem,
                                                  f32x16 (&acc1)[BM / (32 * WM)][BN / (32 * WN)],
                                                  f32x16 (&acc2)[BM / (32 * WM)][BN / (32 * WN)]) {
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int NWAVE = WM * WN;
    constexpr int STAGE = big_stage_halves<BM, BN>();
    constexpr int PA = BM / 16, PW = BN / 16;           // 16-row DMA pieces per plane
    static_assert(PA % NWAVE == 0 && PW % NWAVE == 0, "plane boundaries fall on multiples of the wave count");
    constexpr int JA = PA / NWAVE, JW = PW / NWAVE;     // pieces per wave per A plane / per W plane
    constexpr int NPW = 2 * JA + 2 * JW;                // DMA instructions per wave per k-tile
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nkt = K / BBK;

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc1[i][j][r] = 0.0f; acc2[i][j][r] = 0.0f; }

    // ---- producer side: this wave's NPW pieces of a k-tile.  Piece j of wave w is 16-row group
    // w + NWAVE j of the list [A hi | A lo | W hi | W lo]; PA and PW are multiples of NWAVE, so the plane
    // of piece j is a compile-time property and the issue path has no branches.
    const int prow = lane >> 2, pc = lane & 3;
    const int psw = (pc ^ ((prow >> 2) & 3)) * 8;        // logical chunk (halves) this lane fetches
    int aoff[JA], woff[JW];                              // per-lane source offsets (halves) inside a k-panel
#pragma unroll
    for (int q = 0; q < JA; ++q) aoff[q] = min(row0 + (wave + NWAVE * q) * 16 + prow, row_cap - 1) * PANEL_K + psw;
#pragma unroll
    for (int q = 0; q < JW; ++q) woff[q] = min(col0 + (wave + NWAVE * q) * 16 + prow, col_cap - 1) * PANEL_K + psw;
    auto issue = [&](int kt, int stage) {
        const int k = kt * BBK;
        const bool first = k < ga.K0;
        const int ka = first ? k : k - ga.K0;
        const size_t apan = panel_base(ka, a_rows);
        const _Float16* pah = (first ? ga.A0.hi : ga.A1.hi) + apan;
        const _Float16* pal = (first ? ga.A0.lo : ga.A1.lo) + apan;
        const size_t wpan = panel_base(k, col_cap);
        const _Float16* pwh = W.hi + wpan;
        const _Float16* pwl = W.lo + wpan;
        _Float16* sbase = smem + (size_t)stage * STAGE + wave * 16 * BBK;
#pragma unroll
        for (int j = 0; j < JA; ++j) glds16_(pah + aoff[j], sbase + j * NWAVE * 16 * BBK);
#pragma unroll
        for (int j = 0; j < JA; ++j) glds16_(pal + aoff[j], sbase + BM * BBK + j * NWAVE * 16 * BBK);
#pragma unroll
        for (int j = 0; j < JW; ++j) glds16_(pwh + woff[j], sbase + 2 * BM * BBK + j * NWAVE * 16 * BBK);
#pragma unroll
        for (int j = 0; j < JW; ++j) glds16_(pwl + woff[j], sbase + 2 * BM * BBK + BN * BBK + j * NWAVE * 16 * BBK);
    };

    // ---- consumer side: fragment offsets (halves) inside a plane image, per k16 step
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, lr = lane & 31;
    const int fsw = (lr >> 2) & 3;
    const int fo0 = lr * BBK + ((0 + h) ^ fsw) * 8, fo1 = lr * BBK + ((2 + h) ^ fsw) * 8;
    const int abase = wm * 32 * TM * BBK, wbase = 2 * BM * BBK + wn * 32 * TN * BBK;

    // lock-step form: [wait tile kt] barrier [issue tile kt+1 into the stage read before the barrier] [read + MFMA tile kt]
    issue(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // tile kt landed for every wave; tile kt-1 fully read
        if (kt + 1 < nkt) issue(kt + 1, (kt + 1) & 1);
        const _Float16* st = smem + (size_t)(kt & 1) * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int fo = s ? fo1 : fo0;
            half8 fah[TM], fal[TM], fwh[TN], fwl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                fah[i] = *reinterpret_cast<const half8*>(st + abase + i * 32 * BBK + fo);
                fal[i] = *reinterpret_cast<const half8*>(st + abase + BM * BBK + i * 32 * BBK + fo);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                fwh[j] = *reinterpret_cast<const half8*>(st + wbase + j * 32 * BBK + fo);
                fwl[j] = *reinterpret_cast<const half8*>(st + wbase + BN * BBK + j * 32 * BBK + fo);
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
    }
    (void)NPW;
}

}  // namespace sslam
