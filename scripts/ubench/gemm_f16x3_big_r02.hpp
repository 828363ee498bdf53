// r02 copy of csrc/gemm_f16x3_big.hpp with every main-loop form and the ablation switches (ubench only).
// gemm_f16x3_big.hpp - split-precision matrix-core GEMM main loop for BATCHED token sets (gfx950).
//
// The 64-row ring GEMM of gemm_f16x3.hpp is sized for one pair (4096 token rows -> 256 workgroups):
// its 64x128 tile takes in 4 (BM + BN) = 768 bytes of operand planes per k for 2 BM BN = 16 k flop, and
// a CU's LDS-DMA intake (~68 GB/s from L2) - not its matrix pipe - sets the time.  With a batch of
// pairs there are >= 16 k rows, so the tile can grow: 128 x 256 takes in 1536 B per k for 65 k flop,
// twice the intensity, which puts the load path and the three-MFMA-per-product matrix work in balance
//     load   4 (128 + 256) B / k  /  ~34 B/clk  = 45 clk per k
//     MFMA   2 * 128 * 256 * 3 / 4096 flop/clk  = 48 clk per k.
//
// Structure (one workgroup = 8 waves = 2 per SIMD, every wave both loads and computes):
//   * wave (wm, wn) of a WM x WN grid owns a (BM/WM) x (BN/WN) = 64 x 64 sub-tile: 2 x 2 MFMA tiles
//     of 32 x 32, two fp32 accumulators each (hi.hi and the cross terms) = 128 accumulator registers
//   * k-tile = 32 halves (64 B per row per plane): 3-stage LDS ring of (2 BM + 2 BN) x 64 B = 48 KB
//     per stage, filled by LDS-DMA (global_load_lds_dwordx4: 1 KiB = 16 rows x 64 B per wave
//     instruction, 6 per wave per k-tile), two k-tiles in flight behind a counted s_waitcnt vmcnt
//   * ONE raw s_barrier per k-tile: [wait own pieces of tile kt] barrier [refill the stage read
//     before the barrier] [ds_read_b128 fragments + 24 MFMA of tile kt]
//   * LDS image un-padded (DMA writes linearly); the 16-byte chunk index is XOR-swizzled with
//     (row >> 2) & 3 on the SOURCE address and on every fragment read: the 16 lanes of a
//     ds_read_b128 group then touch 16 distinct 16-byte slots of the 256-byte bank row.
#pragma once
#include <type_traits>
#include "gemm_f16x3.hpp"

#ifndef GEMM_ABL
#define GEMM_ABL 0       // ubench ablations: 1 no MFMA, 2 no DMA in the loop, 4 no fragment reads
#endif

namespace sslam {

constexpr int BBK = 32;                       // halves of k per stage
constexpr int BIG_STAGES = 3;

template <int BM, int BN>
constexpr int big_stage_halves() { return (2 * BM + 2 * BN) * BBK; }

// offset (halves) of row 0 of the 32-deep k-tile that starts at column k (k % 32 == 0) of a plane
__device__ __forceinline__ size_t panel_base(int k, int rows_total) {
    return (size_t)(k / PANEL_K) * rows_total * PANEL_K + (k % PANEL_K);
}

// A planes: k-panel layout over `a_rows` rows; W planes: k-panel layout over `col_cap` rows.
// row0/row_cap, col0/col_cap in plane-row units.  K % 32 == 0, ga.K0 % PANEL_K == 0.
template <int BM, int BN, int WM, int WN, int VARIANT = 1, int NSTAGE = BIG_STAGES>
__device__ __forceinline__ void gemm_mainloop_big(const GemmAH& ga, SplitPtr W, int a_rows, int K, int row0,
                                                  int row_cap, int col0, int col_cap, _Float16* smem,
                                                  f32x16 (&acc1)[BM / (32 * WM)][BN / (32 * WN)],
                                                  f32x16 (&acc2)[BM / (32 * WM)][BN / (32 * WN)]) {
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int NWAVE = WM * WN;
    static_assert(NWAVE == 8 || NWAVE == 4, "4 or 8 waves");
    static_assert(VARIANT == 0 || NSTAGE == 3, "the pipelined forms use a 3-stage ring");
    constexpr int STAGE = big_stage_halves<BM, BN>();
    constexpr int PA = BM / 16, PW = BN / 16;           // 16-row DMA pieces per plane
    constexpr int NPIECE = 2 * PA + 2 * PW;
    static_assert(NPIECE % NWAVE == 0, "pieces divide over the waves");
    constexpr int NPW = NPIECE / NWAVE;                 // DMA instructions per wave per k-tile
    static_assert(VARIANT == 0 || PA % NWAVE == 0, "a short A tile (2 PA = NWAVE) only in the lock-step form");
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
    // g = w + NWAVE j of the list [A hi | A lo | W hi | W lo]; PA and PW are multiples of NWAVE, so
    // the plane of piece j is a compile-time property and the issue path has no branches (the
    // scheduler can then place each DMA instruction between MFMAs).
    // (SHORT_A: a 64-row A tile has only 2 PA = NWAVE pieces - wave w takes piece w of [A hi | A lo],
    // a wave-uniform plane choice - and the W planes divide over the waves as usual)
    constexpr bool SHORT_A = (PA % NWAVE) != 0;
    static_assert(SHORT_A ? (2 * PA == NWAVE && PW % NWAVE == 0) : (PA % NWAVE == 0 && PW % NWAVE == 0),
                  "plane boundaries fall on multiples of the wave count");
    constexpr int JA = SHORT_A ? 1 : PA / NWAVE, JW = PW / NWAVE;     // pieces per wave per A plane / per W plane
    const int prow = lane >> 2, pc = lane & 3;
    const int psw = (pc ^ ((prow >> 2) & 3)) * 8;        // logical chunk (halves) this lane fetches
    int aoff[JA], woff[JW];                              // per-lane source offsets (halves) inside a k-panel
#pragma unroll
    for (int q = 0; q < JA; ++q)
        aoff[q] = min(row0 + (SHORT_A ? wave % PA : wave + NWAVE * q) * 16 + prow, row_cap - 1) * PANEL_K + psw;
#pragma unroll
    for (int q = 0; q < JW; ++q) woff[q] = min(col0 + (wave + NWAVE * q) * 16 + prow, col_cap - 1) * PANEL_K + psw;
    const bool lo_plane = SHORT_A && wave >= PA;        // (short A tile: this wave's A piece is of the lo plane)
    const _Float16* const a0w = lo_plane ? ga.A0.lo : ga.A0.hi;
    const _Float16* const a1w = lo_plane ? ga.A1.lo : ga.A1.hi;
    auto issue = [&](int kt, int stage, int j0 = 0, int j1 = 1 << 20) {
        if constexpr (SHORT_A) {
            const int k = kt * BBK;
            const bool first = k < ga.K0;
            const int ka = first ? k : k - ga.K0;
            const size_t apan = panel_base(ka, a_rows);
            const _Float16* pa = (first ? a0w : a1w) + apan;
            const size_t wpan = panel_base(k, col_cap);
            _Float16* st = smem + (size_t)stage * STAGE;
            glds16_(pa + aoff[0], st + (lo_plane ? BM * BBK : 0) + (wave % PA) * 16 * BBK);
            _Float16* wb = st + 2 * BM * BBK + wave * 16 * BBK;
#pragma unroll
            for (int q = 0; q < JW; ++q) glds16_(W.hi + wpan + woff[q], wb + q * NWAVE * 16 * BBK);
#pragma unroll
            for (int q = 0; q < JW; ++q) glds16_(W.lo + wpan + woff[q], wb + BN * BBK + q * NWAVE * 16 * BBK);
            return;
        }
#if GEMM_ABL & 2
        if (kt > 2) return;
#endif
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
        for (int j = 0; j < NPW; ++j) {
            if (j < j0 || j >= j1) continue;
            if (j < JA) glds16_(pah + aoff[j], sbase + j * NWAVE * 16 * BBK);
            else if (j < 2 * JA) glds16_(pal + aoff[j - JA], sbase + BM * BBK + (j - JA) * NWAVE * 16 * BBK);
            else if (j < 2 * JA + JW)
                glds16_(pwh + woff[j - 2 * JA], sbase + 2 * BM * BBK + (j - 2 * JA) * NWAVE * 16 * BBK);
            else
                glds16_(pwl + woff[j - 2 * JA - JW],
                        sbase + 2 * BM * BBK + BN * BBK + (j - 2 * JA - JW) * NWAVE * 16 * BBK);
        }
    };

    // ---- consumer side: fragment offsets (halves) inside a plane image, per k16 step
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, lr = lane & 31;
    const int fsw = (lr >> 2) & 3;
    const int fo0 = lr * BBK + ((0 + h) ^ fsw) * 8, fo1 = lr * BBK + ((2 + h) ^ fsw) * 8;
    const int abase = wm * 32 * TM * BBK, wbase = 2 * BM * BBK + wn * 32 * TN * BBK;

    auto read_frags = [&](const _Float16* st, int fo, half8 (&fah)[TM], half8 (&fal)[TM], half8 (&fwh)[TN],
                          half8 (&fwl)[TN]) {
#if GEMM_ABL & 4
        st = smem; fo = lane * 8;
        if (acc1[0][0][0] != 0.25f) return;
#endif
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
    };
    auto mma = [&](const half8 (&fah)[TM], const half8 (&fal)[TM], const half8 (&fwh)[TN], const half8 (&fwl)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#if GEMM_ABL & 1
                acc1[i][j][0] += (float)fah[i][0] + (float)fwh[j][1];
                acc2[i][j][0] += (float)fal[i][0] + (float)fwl[j][1];
#else
                acc1[i][j] = mfma16(fah[i], fwh[j], acc1[i][j]);
                acc2[i][j] = mfma16(fah[i], fwl[j], acc2[i][j]);
                acc2[i][j] = mfma16(fal[i], fwh[j], acc2[i][j]);
#endif
            }
    };

    if constexpr (VARIANT == 0) {
        // lock-step form: [wait tile kt] barrier [issue tile kt+2] [read + MFMA tile kt]
        issue(0, 0);
        if (NSTAGE > 2 && nkt > 1) issue(1, 1);
        for (int kt = 0; kt < nkt; ++kt) {
            if (NSTAGE > 2 && kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();               // tile kt landed for every wave; tile kt-1 fully read
            if (kt + NSTAGE - 1 < nkt) issue(kt + NSTAGE - 1, (kt + NSTAGE - 1) % NSTAGE);
            const _Float16* st = smem + (size_t)(kt % NSTAGE) * STAGE;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                half8 fah[TM], fal[TM], fwh[TN], fwl[TN];
                read_frags(st, s ? fo1 : fo0, fah, fal, fwh, fwl);
                mma(fah, fal, fwh, fwl);
            }
        }
    } else if constexpr (VARIANT == 3) {
        // as the pipelined form below, with the DMA instructions of a k-tile SPREAD over the MFMAs: all
        // eight waves issuing their pieces in one burst behind the barrier back the texture addresser up
        // and every wave waits in its issue slot (measured: the burst is ~600 cycles during which no
        // MFMA runs).  Here a wave issues one piece per four MFMAs - three of tile kt+2 beside the first
        // k16 step of tile kt, three of tile kt+3 beside the second - and the two waves of a SIMD
        // issue at different MFMA positions.
        static_assert(NPW % 2 == 0, "pieces split over the two k16 steps");
        constexpr int H = NPW / 2;
        half8 ah0[TM], al0[TM], wh0[TN], wl0[TN], ah1[TM], al1[TM], wh1[TN], wl1[TN];
        issue(0, 0);
        if (nkt > 1) issue(1, 1);
        if (nkt > 2) issue(2, 2, 0, H);
        if (nkt > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW + H) : "memory");
        else if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        read_frags(smem, fo0, ah0, al0, wh0, wl0);
        // one k-tile; ISS_A: second half of tile kt+2's pieces, ISS_B: first half of tile kt+3's, NEXT: a
        // tile kt+1 exists.  Compile-time flags keep each half a single basic block (the scheduler
        // cannot move a DMA instruction across a branch).
        auto body = [&](int kt, auto iss_a, auto iss_b, auto next) {
            constexpr bool ISS_A = decltype(iss_a)::value, ISS_B = decltype(iss_b)::value, NEXT = decltype(next)::value;
            const _Float16* st = smem + (size_t)(kt % NSTAGE) * STAGE;
            read_frags(st, fo1, ah1, al1, wh1, wl1);
            if constexpr (ISS_A) issue(kt + 2, (kt + 2) % NSTAGE, H, NPW);
            mma(ah0, al0, wh0, wl0);
            if constexpr (ISS_A) {
                // one piece per 12 / H MFMAs (VMEM group = the global_load_lds and nothing else)
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
                for (int q = 0; q + 1 < H; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 12 / H, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 12 / H - 2, 0);
            }
            if constexpr (NEXT) {
                if constexpr (ISS_A) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                read_frags(smem + (size_t)((kt + 1) % NSTAGE) * STAGE, fo0, ah0, al0, wh0, wl0);
                if constexpr (ISS_B) issue(kt + 3, kt % NSTAGE, 0, H);
            }
            mma(ah1, al1, wh1, wl1);
            if constexpr (ISS_B) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
                for (int q = 0; q + 1 < H; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 12 / H, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 12 / H - 2, 0);
            }
        };
        using T_ = std::true_type; using F_ = std::false_type;
        int kt = 0;
        for (; kt + 3 < nkt; ++kt) body(kt, T_{}, T_{}, T_{});
        if (kt + 2 < nkt) { body(kt, T_{}, F_{}, T_{}); ++kt; }
        if (kt + 1 < nkt) { body(kt, F_{}, F_{}, T_{}); ++kt; }
        body(kt, F_{}, F_{}, F_{});
    } else {
        // software-pipelined form: the fragments of k16 step g+1 are read while step g multiplies, the
        // barrier sits between the two steps of a k-tile, and the DMA of tile kt+3 is issued behind it,
        // in front of 12 MFMAs: the partner wave of the SIMD multiplies while this one issues.
        half8 ah0[TM], al0[TM], wh0[TN], wl0[TN], ah1[TM], al1[TM], wh1[TN], wl1[TN];
        issue(0, 0);
        if (nkt > 1) issue(1, 1);
        if (nkt > 2) issue(2, 2);
        if (nkt > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");
        else if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        read_frags(smem, fo0, ah0, al0, wh0, wl0);
        for (int kt = 0; kt < nkt; ++kt) {
            const _Float16* st = smem + (size_t)(kt % BIG_STAGES) * STAGE;
            read_frags(st, fo1, ah1, al1, wh1, wl1);
            if constexpr (VARIANT == 2) __builtin_amdgcn_s_setprio(1);
            mma(ah0, al0, wh0, wl0);
            if constexpr (VARIANT == 2) __builtin_amdgcn_s_setprio(0);
            if (kt + 1 < nkt) {
                // own pieces of tile kt+1 landed (tile kt+2 may stay in flight); every LDS read of
                // tile kt has returned (the stage is refilled behind the barrier)
                if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (kt + 3 < nkt) issue(kt + 3, kt % BIG_STAGES);
                read_frags(smem + (size_t)((kt + 1) % BIG_STAGES) * STAGE, fo0, ah0, al0, wh0, wl0);
            }
            if constexpr (VARIANT == 2) __builtin_amdgcn_s_setprio(1);
            mma(ah1, al1, wh1, wl1);
            if constexpr (VARIANT == 2) __builtin_amdgcn_s_setprio(0);
        }
    }
}

}  // namespace sslam
