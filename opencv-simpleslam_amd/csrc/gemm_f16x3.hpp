// gemm_f16x3.hpp - split-precision matrix-core GEMM main loop (gfx950 only).
//
// An fp32 operand a is carried as two fp16 planes  a = hi + lo * 2^-11  with
// hi = rne_f16(a), lo = rne_f16((a - hi) * 2^11)  (22 significand bits), and a product is
//     a.b = hi_a.hi_b + (hi_a.lo_b + lo_a.hi_b) * 2^-11          (lo.lo = 2^-22 dropped)
// evaluated as three v_mfma_f32_32x32x16_f16 into two fp32 accumulators (c1, c2).  Relative
// error per product ~2^-22 (fp32: 2^-24), accumulation in fp32 inside the matrix core: the
// result is fp32-grade at 16/3 of the fp32 matrix-core rate.  Operands arrive PRE-SPLIT from the
// producer's epilogue (each element is split once, not once per consuming tile), so the main loop
// has no conversion work: global -> registers -> LDS -> ds_read_b128 fragments -> MFMA.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_f32.hpp"

namespace sslam {

using half8 = _Float16 __attribute__((ext_vector_type(8)));
using half4 = _Float16 __attribute__((ext_vector_type(4)));
using half2v = _Float16 __attribute__((ext_vector_type(2)));

constexpr float SPLIT_SCALE = 2048.0f;          // 2^11
constexpr float SPLIT_INV = 1.0f / 2048.0f;

// RANGE.  A FINITE value with |a| >= 65520 does not fit the fp16 planes (it rounds to +-inf in the high
// plane and the product silently becomes inf / NaN).  Every split reports such a value through the
// `range_flag` word its caller passes - per matcher instance AND per pair (LGCtrl::range_overflow,
// lightglue_kernels.hip): the pair's result is then marked invalid (match count -1) and the host
// entries turn it into an error; the exact-fp32 path (precision 0) has no such limit.  Non-finite
// inputs are not flagged: they stay NaN / inf as in any fp32 evaluation.
__device__ __forceinline__ void split_f32(float a, _Float16& hi, _Float16& lo, int* range_flag) {
    // an fp16 subnormal is not a safe MFMA operand (flushed on input): below 2^-14 the whole
    // value moves into the scaled low plane, which keeps 11 bits down to |a| = 2^-25
    const float aa = fabsf(a);
    if (aa >= 65520.0f && aa < INFINITY) {
        *range_flag = 1;
        hi = (_Float16)copysignf(65504.0f, a);
        lo = (_Float16)fminf(fmaxf((a - (float)hi) * SPLIT_SCALE, -65504.0f), 65504.0f);
        return;
    }
    hi = aa < 6.103515625e-5f ? (_Float16)0.0f : (_Float16)a;
    lo = (_Float16)((a - (float)hi) * SPLIT_SCALE);
}

// Branch-free split of eight values for the GEMM epilogues (64 values per thread and tile: the
// branchy scalar form above is ~37 instructions per value there, ~6 us of VALU per 128 x 128 tile).
// Same planes as split_f32 for every in-range value (hi = rne(a) or 0 below 2^-14, lo = the exact
// residual x 2^11 rounded once; NaN / inf stay non-finite in both planes).  A finite |a| >= 65520 is
// NOT saturated here: it leaves as inf (and poisons what consumes it) but is reported through `amax` -
// the caller keeps the running maximum of |a| and raises the range flag once per thread
// (split_range_check).
// ~5 instructions per value: v_cmp + v_cndmask, 1/2 v_cvt_pk_f16_f32, v_mul, v_fma_mix, 1/2 v_max3.
// two values -> packed (hi, lo) half2 words
__device__ __forceinline__ void split2_fast(float a0, float a1, unsigned& h2, unsigned& l2, float& amax) {
    typedef float float2s __attribute__((ext_vector_type(2)));
    const float neg_scale = -SPLIT_SCALE;
    amax = fmaxf(amax, fmaxf(fabsf(a0), fabsf(a1)));
    const float2s z = {fabsf(a0) < 6.103515625e-5f ? 0.0f : a0, fabsf(a1) < 6.103515625e-5f ? 0.0f : a1};
    h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(z, half2v));
    const float t0 = a0 * SPLIT_SCALE, t1 = a1 * SPLIT_SCALE;
    // l2.lo = fp16(h2.lo * -2^11 + t0), l2.hi = fp16(h2.hi * -2^11 + t1): one rounding each
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l2) : "v"(h2), "v"(neg_scale), "v"(t0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l2) : "v"(h2), "v"(neg_scale), "v"(t1));
}
__device__ __forceinline__ void split8_fast(const float (&v)[8], uint4& hi, uint4& lo, float& amax) {
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split2_fast(v[2 * e], v[2 * e + 1], h[e], l[e], amax);
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ void split_range_check(float amax, int* range_flag) {
    if (amax >= 65520.0f && amax < INFINITY) *range_flag = 1;
}

__device__ __forceinline__ f32x16 mfma16(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// split planes of a [rows][ld] matrix
struct SplitPtr {
    const _Float16* hi;
    const _Float16* lo;
};

constexpr int HBK = 32;              // k per LDS tile (halves)
constexpr int HLD = HBK + 8;         // +16 B pad: 80 B row stride, ds_read_b128 conflict-free

template <int BM, int BN>
struct __attribute__((aligned(16))) GemmSmemH {
    _Float16 a_hi[2][BM * HLD];
    _Float16 a_lo[2][BM * HLD];
    _Float16 w_hi[2][BN * HLD];
    _Float16 w_lo[2][BN * HLD];
};

struct GemmAH {                      // A operand: optional concat of two split sources along K
    SplitPtr A0;
    SplitPtr A1;
    int lda;                         // same leading dimension for both
    int K0;                          // columns [0,K0) from A0, [K0,K) from A1 (K0 % 32 == 0)
};

#define LDH8(dst, ptr) dst = *reinterpret_cast<const uint4*>(ptr)
#define STH8(ptr, src) *reinterpret_cast<uint4*>(ptr) = src

// acc1/acc2: hi.hi and cross-term accumulators; result = acc1 + acc2 * 2^-11
template <int BM, int BN, int TM, int TN>
__device__ __forceinline__ void gemm_mainloop_h(const GemmAH& ga, SplitPtr W, int ldw, int K, int row0,
                                                int row_cap, int col0, int col_cap, GemmSmemH<BM, BN>& sm,
                                                f32x16 (&acc1)[TM][TN], f32x16 (&acc2)[TM][TN]) {
    static_assert(BM == 64 * TM && BN == 64 * TN, "2x2 waves of 32*TM x 32*TN");
    constexpr int NA = BM / 64, NW = BN / 64;     // 16-byte loads per thread per plane per k-tile
    static_assert(NA <= 2 && NW <= 2, "named prefetch registers cover up to 128-row tiles");
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, lr = lane & 31;

    // named prefetch registers (see gemm_f32.hpp for why not arrays)
    uint4 ah0, ah1, al0, al1, wh0, wh1, wl0, wl1;
    ah0 = ah1 = al0 = al1 = wh0 = wh1 = wl0 = wl1 = make_uint4(0, 0, 0, 0);
    const int lr4 = t >> 2, lc8 = (t & 3) * 8;    // (row, k-offset in halves) inside a 64-row slab
    size_t oa[2], ow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        oa[j] = (size_t)min(row0 + lr4 + 64 * j, row_cap - 1) * ga.lda + lc8;
        ow[j] = (size_t)min(col0 + lr4 + 64 * j, col_cap - 1) * ldw + lc8;
    }
    const int sto = lr4 * HLD + lc8;

#define GEMMH_GLOAD(kt_)                                                          \
    {                                                                             \
        const int k_ = (kt_) * HBK;                                               \
        const bool f_ = k_ < ga.K0;                                               \
        const _Float16* pah_ = f_ ? ga.A0.hi + k_ : ga.A1.hi + (k_ - ga.K0);      \
        const _Float16* pal_ = f_ ? ga.A0.lo + k_ : ga.A1.lo + (k_ - ga.K0);      \
        LDH8(ah0, pah_ + oa[0]); LDH8(al0, pal_ + oa[0]);                         \
        if constexpr (NA > 1) { LDH8(ah1, pah_ + oa[1]); LDH8(al1, pal_ + oa[1]); } \
        LDH8(wh0, W.hi + k_ + ow[0]); LDH8(wl0, W.lo + k_ + ow[0]);               \
        if constexpr (NW > 1) { LDH8(wh1, W.hi + k_ + ow[1]); LDH8(wl1, W.lo + k_ + ow[1]); } \
    }
#define GEMMH_SSTORE(buf_)                                                        \
    {                                                                             \
        STH8(&sm.a_hi[buf_][sto], ah0); STH8(&sm.a_lo[buf_][sto], al0);           \
        if constexpr (NA > 1) { STH8(&sm.a_hi[buf_][sto + 64 * HLD], ah1); STH8(&sm.a_lo[buf_][sto + 64 * HLD], al1); } \
        STH8(&sm.w_hi[buf_][sto], wh0); STH8(&sm.w_lo[buf_][sto], wl0);           \
        if constexpr (NW > 1) { STH8(&sm.w_hi[buf_][sto + 64 * HLD], wh1); STH8(&sm.w_lo[buf_][sto + 64 * HLD], wl1); } \
    }

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc1[i][j][r] = 0.0f; acc2[i][j][r] = 0.0f; }

    const int nkt = K / HBK;
    GEMMH_GLOAD(0);
    GEMMH_SSTORE(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) GEMMH_GLOAD(kt + 1);
        const int ra = (wm * 32 * TM + lr) * HLD + 8 * h;
        const int rw = (wn * 32 * TN + lr) * HLD + 8 * h;
#pragma unroll
        for (int st = 0; st < HBK / 16; ++st) {
            half8 fah[TM], fal[TM], fwh[TN], fwl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                fah[i] = *reinterpret_cast<const half8*>(&sm.a_hi[cur][ra + i * 32 * HLD + st * 16]);
                fal[i] = *reinterpret_cast<const half8*>(&sm.a_lo[cur][ra + i * 32 * HLD + st * 16]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                fwh[j] = *reinterpret_cast<const half8*>(&sm.w_hi[cur][rw + j * 32 * HLD + st * 16]);
                fwl[j] = *reinterpret_cast<const half8*>(&sm.w_lo[cur][rw + j * 32 * HLD + st * 16]);
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
        if (kt + 1 < nkt) GEMMH_SSTORE(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#undef GEMMH_GLOAD
#undef GEMMH_SSTORE
}


// ------------------------------------------------------------------------------------------
// K-PANEL LAYOUT of every split plane that feeds a GEMM: plane[k / PANEL_K][row][k % PANEL_K].
// PANEL_K = 32 halves = the k depth of the batched GEMM's LDS tile (gemm_f16x3_big.hpp), so the
// 16 rows x 64 B a DMA wave-instruction fetches are ONE contiguous 1 KiB run of whole 128-byte
// lines.  (r01/r02 used 64-wide panels: a 32-deep k-tile then took HALF of every 128-byte line and
// the other half one tile later - every line crossed L2 -> L1 twice, and the per-CU LDS-DMA intake,
// which bounds these loops, carried 50 % useful bytes.)  The 64-deep ring GEMM of the single-pair
// path reads two adjacent panels per k-tile (2 x 512 B runs per DMA instruction).
#ifndef SSLAM_PANEL_K
#define SSLAM_PANEL_K 32
#endif
constexpr int PANEL_K = SSLAM_PANEL_K;
static_assert(PANEL_K == 32 || PANEL_K == 64, "k-panel width");
__host__ __device__ __forceinline__ size_t panel_index(int row, int col, int rows_total) {
    return ((size_t)(col / PANEL_K) * rows_total + row) * PANEL_K + (col % PANEL_K);
}
// offset (halves) of 16-byte chunk `c8` (0..7) of `row` inside the 64-deep k-tile `kt64`
__device__ __forceinline__ size_t panel_tile64_offset(int kt64, int row, int c8, int rows_total) {
    if constexpr (PANEL_K == 64) return ((size_t)kt64 * rows_total + row) * 64 + c8 * 8;
    else return ((size_t)(2 * kt64 + (c8 >> 2)) * rows_total + row) * 32 + (c8 & 3) * 8;
}

// LDS-DMA ring version.  BK = 64 halves (128 B rows), NSTAGE-deep ring in dynamic LDS filled by
// global_load_lds_dwordx4 (no staging registers), tiles stay in flight across raw s_barriers
// behind a counted s_waitcnt vmcnt.  The LDS image is un-padded (a DMA wave-instruction writes
// 8 rows x 128 B linearly); the 16-byte chunk index is XOR-swizzled with (row >> 1) & 7 on the
// SOURCE address and on every fragment read, which makes the ds_read_b128 fragments conflict-free.
// ------------------------------------------------------------------------------------------
constexpr int RBK = 64;

template <int BM, int BN>
constexpr int ring_stage_halves() { return (2 * BM + 2 * BN) * RBK; }

__device__ __forceinline__ void glds16_(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// A planes: panel layout over `a_rows` rows (ga.lda is unused); W planes: panel layout over
// `col_cap` rows.  row0/row_cap, col0/col_cap are in plane-row units.
//
// WAVE SPECIALISATION.  The block has 8 waves: waves 0-3 only consume (ds_read + MFMA, 2x2 tile
// layout), waves 4-7 only produce (LDS-DMA).  Issuing one 1 KiB DMA piece costs a wave ~100-200
// cycles, and a k-tile is 32-64 pieces: issued by the consumers themselves that is 2-3x the MFMA
// time of the tile and sits on their critical path (measured: 1.8 us per k-tile); on dedicated
// producer waves it overlaps the consumers' work.  One s_barrier per k-tile, joined by all 8 waves:
//   producer: wait (counted vmcnt) until tile kt has landed -> barrier -> refill the stage the
//             consumers finished reading before that barrier
//   consumer: barrier -> read tile kt
template <int BM, int BN, int TM, int TN, int NSTAGE>
__device__ __forceinline__ void gemm_mainloop_ring(const GemmAH& ga, SplitPtr W, int a_rows, int K, int row0,
                                                   int row_cap, int col0, int col_cap, _Float16* smem,
                                                   f32x16 (&acc1)[TM][TN], f32x16 (&acc2)[TM][TN]) {
    static_assert(BM == 64 * TM && BN == 64 * TN, "2x2 consumer waves of 32*TM x 32*TN");
    constexpr int STAGE = ring_stage_halves<BM, BN>();
    constexpr int GA = BM / 8, GW = BN / 8;            // 8-row DMA groups per plane
    constexpr int NI = (2 * GA + 2 * GW) / 4;          // DMA instructions per producer wave per tile
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool producer = wave >= 4;
    const int nkt = K / RBK;

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc1[i][j][r] = 0.0f; acc2[i][j][r] = 0.0f; }

    if (producer) {
        const int pw = wave - 4;
        const int lrow = lane >> 3, lcp = lane & 7;
        // stage layout (halves): [A_hi BM*64][A_lo BM*64][W_hi BN*64][W_lo BN*64]
        auto issue = [&](int kt, int stage) {
            const int k = kt * RBK;
            const bool first = k < ga.K0;
            const int akt = first ? kt : kt - ga.K0 / RBK;
            const _Float16* pah = first ? ga.A0.hi : ga.A1.hi;
            const _Float16* pal = first ? ga.A0.lo : ga.A1.lo;
            _Float16* sbase = smem + (size_t)stage * STAGE;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int g = pw + 4 * j;                  // wave-uniform group id
                int plane, grp;
                if (g < GA) { plane = 0; grp = g; }
                else if (g < 2 * GA) { plane = 1; grp = g - GA; }
                else if (g < 2 * GA + GW) { plane = 2; grp = g - 2 * GA; }
                else { plane = 3; grp = g - 2 * GA - GW; }
                const int row = grp * 8 + lrow;
                const int c = lcp ^ ((row >> 1) & 7);
                const _Float16* src;
                _Float16* dst;
                if (plane < 2) {
                    src = (plane == 0 ? pah : pal) + sslam::panel_tile64_offset(akt, min(row0 + row, row_cap - 1), c, a_rows);
                    dst = sbase + plane * BM * RBK + grp * 8 * RBK;
                } else {
                    src = (plane == 2 ? W.hi : W.lo) + sslam::panel_tile64_offset(kt, min(col0 + row, col_cap - 1), c, col_cap);
                    dst = sbase + 2 * BM * RBK + (plane - 2) * BN * RBK + grp * 8 * RBK;
                }
                glds16_(src, dst);
            }
        };
#pragma unroll
        for (int pre = 0; pre < NSTAGE - 1; ++pre)
            if (pre < nkt) issue(pre, pre);
        for (int kt = 0; kt < nkt; ++kt) {
            // tile kt has landed once at most the (NSTAGE-2) younger tiles of this wave are outstanding
            const int younger = min(nkt - 1 - kt, NSTAGE - 2);
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + NSTAGE - 1 < nkt) issue(kt + NSTAGE - 1, (kt + NSTAGE - 1) % NSTAGE);
        }
        return;
    }

    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, lr = lane & 31;
    for (int kt = 0; kt < nkt; ++kt) {
        __builtin_amdgcn_s_barrier();                  // tile kt is in LDS (all producers waited for it)
        const _Float16* st = smem + (size_t)(kt % NSTAGE) * STAGE;
        const _Float16 *sah = st, *sal = st + BM * RBK, *swh = st + 2 * BM * RBK, *swl = st + 2 * BM * RBK + BN * RBK;
#pragma unroll
        for (int s = 0; s < RBK / 16; ++s) {
            half8 fah[TM], fal[TM], fwh[TN], fwl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * 32 * TM + i * 32 + lr;
                const int o = row * RBK + (((2 * s + h) ^ ((row >> 1) & 7)) * 8);
                fah[i] = *reinterpret_cast<const half8*>(sah + o);
                fal[i] = *reinterpret_cast<const half8*>(sal + o);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * 32 * TN + j * 32 + lr;
                const int o = row * RBK + (((2 * s + h) ^ ((row >> 1) & 7)) * 8);
                fwh[j] = *reinterpret_cast<const half8*>(swh + o);
                fwl[j] = *reinterpret_cast<const half8*>(swl + o);
            }
#if !defined(SSLAM_DBG_NOMFMA)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc1[i][j] = mfma16(fah[i], fwh[j], acc1[i][j]);
                    acc2[i][j] = mfma16(fah[i], fwl[j], acc2[i][j]);
                    acc2[i][j] = mfma16(fal[i], fwh[j], acc2[i][j]);
                }
#else
            acc1[0][0][0] += (float)fah[0][0] + (float)fal[0][1] + (float)fwh[0][2] + (float)fwl[0][3];
#endif
        }
    }
}

}  // namespace sslam
