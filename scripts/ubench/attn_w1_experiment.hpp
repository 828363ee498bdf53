// attn_w1_experiment.hpp - the one-wave-per-SIMD attention experiment (r02), kept OUT of the product:
// included by scripts/ubench/attn_bench.hip after the product translation unit (it uses its internals).
// Bit-compatible with lg_attention_p_kernel up to fp32 rounding (ATTN_PP=2 ATTN_CMP=1), slower on real
// data (305 vs 231 us per 8-pair launch on random operands, 189 vs 180 on zeros): see
// profiles/r02_attention_experiments.md.  Build with -DW1_PADN=3: the asm MFMAs need two wait states
// after the v_accvgpr moves the register allocator puts in front of them (the compiler does not see
// an MFMA inside asm and inserts none).
#pragma once
namespace {
// ---- attention, split precision, ONE WAVE PER SIMD, hand-placed gaps (batched launches; experiment) ----
// A wave owns the whole SIMD (512 registers) and 64 queries as two independent 32-query blocks X and Y;
// its instruction stream alternates
//   QK phase: S_X(j), S_Y(j) = K_j Q^T      (24 MFMA)   with the softmax of Y's sub-step j-1 in its gaps
//   PV phase: O_X, O_Y += V^T_{j-1} P(j-1)  (24 MFMA)   with the softmax of X's sub-step j   in its gaps
// one MFMA + one softmax slice + one fragment read or DMA piece per gap, pinned gap by gap with
// sched_barrier(0).  asm MFMAs fix the register files: O, Q and the K fragments in AGPRs (ds_read lands
// them there directly), logits, P and the V^T fragments in arch VGPRs.  K(t+1) and V^T(t) are exactly
// what is read between the barrier in the middle of tile t and the next one: one barrier per tile.
struct W1Softmax {
    float sv[16];
    float m_run, l_run, mb, alpha, tmax, ps0, ps1;
    bool rescale;
};

template <bool MASK>
__device__ __forceinline__ void w1_slice(int g, W1Softmax& st, const f32x16& s1, const f32x16& s2, int kbase, int nk,
                                         int lane, bool qvalid, unsigned (&hu)[8], unsigned (&lu)[8]) {
#ifndef W1_SKIP
#define W1_SKIP 0       // ubench ablation: 1 combine, 2 max / decision, 4 exp, 8 split, 16 row sums, 32 fragment reads, 64 DMA
#endif
    if ((W1_SKIP & 1) && g < 4) return;
    if ((W1_SKIP & 2) && g >= 4 && g < 8) return;
    if ((W1_SKIP & 4) && g >= 8 && g < 16) return;
    if ((W1_SKIP & 8) && g >= 16 && g < 20) return;
    if ((W1_SKIP & 16) && g >= 20) return;
    if (g < 4) {
#pragma unroll
        for (int i = 4 * g; i < 4 * g + 4; ++i) {
            st.sv[i] = __builtin_fmaf(s2[i], SPLIT_INV, s1[i]);
            if constexpr (MASK) { if (kbase + acc_row(i, lane) >= nk) st.sv[i] = -INFINITY; }
        }
    } else if (g == 4) {
        st.tmax = fmaxf(fmaxf(st.sv[0], st.sv[1]), st.sv[2]);
        st.ps0 = fmaxf(fmaxf(st.sv[3], st.sv[4]), st.sv[5]);
        st.ps1 = fmaxf(fmaxf(st.sv[6], st.sv[7]), st.sv[8]);
    } else if (g == 5) {
        st.tmax = fmaxf(fmaxf(st.tmax, st.ps0), st.ps1);
        st.ps0 = fmaxf(fmaxf(st.sv[9], st.sv[10]), st.sv[11]);
        st.ps1 = fmaxf(fmaxf(st.sv[12], st.sv[13]), st.sv[14]);
    } else if (g == 6) {
        st.tmax = fmaxf(fmaxf(st.tmax, st.ps0), fmaxf(st.ps1, st.sv[15]));
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(st.tmax), __float_as_uint(st.tmax), false, false);
        st.tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    } else if (g == 7) {
        // PER-LANE deferred rescale (the two halves of a query see the same tmax, so they agree): the
        // 4-wave kernel's wave-uniform vote puts a VALU -> SALU -> VALU round trip (v_cmp, s_and, s_cmp,
        // s_cselect, v_cndmask) on the path to the exponentials - ~500 cycles per phase with no second
        // wave to cover it.  Only the O-rescale BRANCH needs the wave-wide answer, and not before the
        // end of the phase.  (m then moves for fewer lanes than in the 4-wave kernel: same softmax,
        // different - equally valid - reference, results equal to fp32 rounding, not bit for bit.)
        const bool up = st.tmax > st.m_run + RESCALE_THR;
        const float m_new = up ? fmaxf(st.m_run, st.tmax) : st.m_run;
        st.alpha = __builtin_amdgcn_exp2f(st.m_run - m_new);     // 1 when the reference stays; 0 on the first sub-step
        st.rescale = __any(qvalid && up);
        st.m_run = m_new;
        st.mb = fabsf(m_new) < 4.0e6f ? m_new - P_BIAS : m_new;
        st.ps0 = 0.0f; st.ps1 = 0.0f;
        if (W1_SKIP & 128) st.rescale = false;      // timing experiment: the decision is computed, O is never rescaled
    } else if (g < 16) {
        const int k = g - 8;
        const float d0 = st.sv[2 * k] - st.mb, d1 = st.sv[2 * k + 1] - st.mb;
        st.sv[2 * k] = __builtin_amdgcn_exp2f(d0);
        st.sv[2 * k + 1] = __builtin_amdgcn_exp2f(d1);
    } else if (g < 20) {
        typedef float float2w __attribute__((ext_vector_type(2)));
        const int q0 = 2 * (g - 16), q1 = q0 + 1;
        const float2w a2 = {st.sv[2 * q0], st.sv[2 * q0 + 1]}, b2 = {st.sv[2 * q1], st.sv[2 * q1 + 1]};
        const unsigned ha = __builtin_bit_cast(unsigned, __builtin_convertvector(a2, sslam::half2v));
        const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(b2, sslam::half2v));
        unsigned la, lb;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(la) : "v"(ha), "v"(st.sv[2 * q0]));
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hb), "v"(st.sv[2 * q1]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(la) : "v"(ha), "v"(st.sv[2 * q0 + 1]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lb) : "v"(hb), "v"(st.sv[2 * q1 + 1]));
        hu[q0] = ha; lu[q0] = la; hu[q1] = hb; lu[q1] = lb;
    } else {
        const int k = g - 20;
        st.ps0 += st.sv[4 * k]; st.ps1 += st.sv[4 * k + 1];
        st.ps0 += st.sv[4 * k + 2]; st.ps1 += st.sv[4 * k + 3];
        if (g == 23) st.l_run = st.l_run * st.alpha + (st.ps0 + st.ps1);
    }
}

__global__ __launch_bounds__(256, 1) void lg_attention_w1_kernel(AttnArgsH p) {
    __shared__ AttnSmemH sm;
    const int nqb = gridDim.x, nslab = gridDim.y;
    int slab, qb;
    {
        const int b = blockIdx.y * gridDim.x + blockIdx.x;
        if ((nslab & 7) == 0) { const int xcd = b & 7, idx = b >> 3; slab = xcd + 8 * (idx / nqb); qb = idx % nqb; }
        else { slab = blockIdx.y; qb = blockIdx.x; }
    }
#ifdef ATTN_BATCH_EMU
    const int ih = slab & 7;
#else
    const int ih = slab;
#endif
    const int img = ih >> 2, head = ih & 3;
    if (ctrl_of(p.ctrl, img).stop) return;
    const int kimg = p.cross ? (img ^ 1) : img;
    const int nq = n_of(p.ctrl, img), nk = n_of(p.ctrl, kimg);
    const int q0 = qb * AQ2;
    if (q0 >= nq) return;
    const int t = threadIdx.x, lane = t & 63, h = lane >> 5, lr = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int T = (nk + AK - 1) / AK;

    const size_t qoff = ((size_t)img * NH + head) * p.Kc * DH;
    const size_t koff = ((size_t)kimg * NH + head) * p.Kc * DH;
    const int qrowX = q0 + wave * 64 + lr, qrowY = qrowX + 32;
    const bool qvX = qrowX < nq, qvY = qrowY < nq;
    half8 qhX[4], qlX[4], qhY[4], qlY[4];
    {
        const int qiX = min(qrowX, p.Kc - 1), qiY = min(qrowY, p.Kc - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qhX[s] = *reinterpret_cast<const half8*>(p.Q.hi + qoff + (size_t)qiX * DH + 16 * s + 8 * h);
            qlX[s] = *reinterpret_cast<const half8*>(p.Q.lo + qoff + (size_t)qiX * DH + 16 * s + 8 * h);
            qhY[s] = *reinterpret_cast<const half8*>(p.Q.hi + qoff + (size_t)qiY * DH + 16 * s + 8 * h);
            qlY[s] = *reinterpret_cast<const half8*>(p.Q.lo + qoff + (size_t)qiY * DH + 16 * s + 8 * h);
        }
    }
    // wave w owns plane w of a tile (K hi, K lo, V^T hi, V^T lo): 8 DMA pieces of 8 rows; the LDS image is
    // [k_hi[2] | k_lo[2] | vt_hi[2] | vt_lo[2]] x 4096 halves, so plane w / buffer b starts at (2 w + b) * 4096
    const _Float16* gplane = (wave == 0 ? p.K.hi : wave == 1 ? p.K.lo : wave == 2 ? p.VT.hi : p.VT.lo) + koff;
    const bool is_v = wave >= 2;
    _Float16* const lds0 = reinterpret_cast<_Float16*>(&sm) + wave * 2 * (AK * DH);
    const int lrow = lane >> 3, lcp = lane & 7;
    auto issue_piece = [&](int tile, int buf, int rg) {
        const int row = rg * 8 + lrow;
        const int c = lcp ^ ((row >> 1) & 7);
        // (V^T has Kc / 64 whole tiles, so the clamp only ever acts on K rows past Kc)
        const _Float16* src = gplane + (size_t)min(tile * AK + row, p.Kc - 1) * DH + c * 8;
        glds16(src, lds0 + buf * (AK * DH) + rg * 8 * DH);
    };

    int koffs[2][4], voffs[2][2][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        const int krow = sub * 32 + lr, kswz = (krow >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; ++s) koffs[sub][s] = krow * DH + (((2 * s + h) ^ kswz) * 8);
#pragma unroll
        for (int s2i = 0; s2i < 2; ++s2i)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int d = db * 32 + lr, vswz = (d >> 1) & 7, c0 = 4 * sub + 2 * s2i + h;
                voffs[sub][s2i][db] = d * AK + ((c0 ^ vswz) * 8);
            }
    }

    f32x16 oX1a, oX2a, oX1b, oX2b, oY1a, oY2a, oY1b, oY2b, sX1, sX2, sY1, sY2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        oX1a[r] = 0.0f; oX2a[r] = 0.0f; oX1b[r] = 0.0f; oX2b[r] = 0.0f;
        oY1a[r] = 0.0f; oY2a[r] = 0.0f; oY1b[r] = 0.0f; oY2b[r] = 0.0f;
        sX1[r] = 0.0f; sX2[r] = 0.0f; sY1[r] = 0.0f; sY2[r] = 0.0f;
    }
    W1Softmax stX, stY;
    stX.m_run = -INFINITY; stX.l_run = 0.0f; stX.alpha = 1.0f; stX.rescale = false;
    stY.m_run = -INFINITY; stY.l_run = 0.0f; stY.alpha = 1.0f; stY.rescale = false;
    unsigned huX[8], luX[8], huY[8], luY[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { huX[i] = 0u; luX[i] = 0u; huY[i] = 0u; luY[i] = 0u; }
    half8 kfh[4], kfl[4], vfh[2][2], vfl[2][2];
    typedef unsigned uint4w __attribute__((ext_vector_type(4)));
    auto pfrag = [](const unsigned (&u)[8], int i) {
        const uint4w v = {u[4 * i], u[4 * i + 1], u[4 * i + 2], u[4 * i + 3]};
        return __builtin_bit_cast(half8, v);
    };
    auto load_k = [&](int buf, int sub, int i) {
        const int s = i >> 1;
        if (i & 1) kfl[s] = *reinterpret_cast<const half8*>(&sm.k_lo[buf][koffs[sub][s]]);
        else       kfh[s] = *reinterpret_cast<const half8*>(&sm.k_hi[buf][koffs[sub][s]]);
    };
    auto load_v = [&](int buf, int sub, int i) {
        const int s2i = i >> 2, db = (i >> 1) & 1;
        if (i & 1) vfl[s2i][db] = *reinterpret_cast<const half8*>(&sm.vt_lo[buf][voffs[sub][s2i][db]]);
        else       vfh[s2i][db] = *reinterpret_cast<const half8*>(&sm.vt_hi[buf][voffs[sub][s2i][db]]);
    };
#define W1_GAP() __builtin_amdgcn_sched_barrier(0)
    auto qk_mfma = [&](int g) {
        constexpr int BLK[24] = {0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 1, 1, 1, 1, 1};
        constexpr int STP[24] = {0, 0, 0, 1, 1, 1, 0, 2, 0, 2, 0, 2, 1, 3, 1, 3, 1, 3, 2, 2, 2, 3, 3, 3};
        constexpr int WHI[24] = {1, 0, 2, 0, 1, 2, 0, 0, 1, 1, 2, 2, 0, 0, 1, 1, 2, 2, 1, 0, 2, 0, 1, 2};
        const int b = BLK[g], s = STP[g], w = WHI[g];
        f32x16& acc = b == 0 ? (w == 0 ? sX1 : sX2) : (w == 0 ? sY1 : sY2);
        const half8 ka = w == 2 ? kfl[s] : kfh[s];
        const half8 qb2 = b == 0 ? (w == 1 ? qlX[s] : qhX[s]) : (w == 1 ? qlY[s] : qhY[s]);
#ifndef W1_PADN
#define W1_PADN 0
#endif
#if W1_PADN == 2
#define W1_PAD "s_nop 15\n\ts_nop 15\n\t"
#elif W1_PADN == 1
#define W1_PAD "s_nop 7\n\t"
#elif W1_PADN == 3
#define W1_PAD "s_nop 1\n\t"
#elif W1_PADN == 4
#define W1_PAD "s_nop 3\n\t"
#else
#define W1_PAD ""
#endif
        if (s == 0 && w < 2) asm volatile(W1_PAD "v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "a"(ka), "a"(qb2));
        else asm volatile(W1_PAD "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(ka), "a"(qb2));
    };
    auto pv_mfma = [&](int g) {
        const int b = g / 12, r = g % 12, i = r / 6, w = r % 6;
        const half8 ph = pfrag(b ? huY : huX, i), pl = pfrag(b ? luY : luX, i);
        f32x16& o1a = b ? oY1a : oX1a; f32x16& o1b = b ? oY1b : oX1b;
        f32x16& o2a = b ? oY2a : oX2a; f32x16& o2b = b ? oY2b : oX2b;
        f32x16& o = (w == 0 || w == 4) ? o1a : (w == 1 || w == 5) ? o1b : w == 2 ? o2a : o2b;
        const half8 va = w == 2 ? vfl[i][0] : w == 3 ? vfl[i][1] : (w & 1) ? vfh[i][1] : vfh[i][0];
        const half8 pb = w >= 4 ? pl : ph;
        asm volatile(W1_PAD "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(o) : "v"(va), "v"(pb));
    };
    auto scale_acc = [](f32x16& o, float alpha) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float e = o[r], tmp;
            asm volatile("v_accvgpr_read_b32 %1, %0\n\ts_nop 0\n\tv_mul_f32 %1, %1, %2\n\ts_nop 0\n\tv_accvgpr_write_b32 %0, %1"
                         : "+a"(e), "=&v"(tmp) : "v"(alpha));
            o[r] = e;
        }
    };
    auto rescale_o = [&](W1Softmax& st, f32x16& a, f32x16& b2, f32x16& c, f32x16& d) {
        if (st.rescale) { scale_acc(a, st.alpha); scale_acc(b2, st.alpha); scale_acc(c, st.alpha); scale_acc(d, st.alpha); }
    };

#pragma unroll
    for (int rg = 0; rg < 8; ++rg) issue_piece(0, 0, rg);
    if (!is_v && T > 1) {
#pragma unroll
        for (int rg = 0; rg < 8; ++rg) issue_piece(1, 1, rg);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) load_k(0, 0, i);

    auto tile_body = [&](auto mask_c, auto first_c, int tile) {
        constexpr bool MASK = decltype(mask_c)::value, FIRST = decltype(first_c)::value;
        const int b = tile & 1;
        const int dma_tile = min(is_v ? tile + 1 : tile + 2, T - 1), dma_buf = is_v ? (b ^ 1) : b;
        W1_GAP();
#pragma unroll
        for (int g = 0; g < 24; ++g) {
            qk_mfma(g);
            if (g < 8 && !(W1_SKIP & 32)) load_v(FIRST ? 0 : b ^ 1, FIRST ? 0 : 1, g);
            if constexpr (!FIRST) w1_slice<false>(g, stY, sY1, sY2, 0, nk, lane, qvY, huY, luY);
            W1_GAP();
        }
        if constexpr (!FIRST) rescale_o(stY, oY1a, oY2a, oY1b, oY2b);
        W1_GAP();
#pragma unroll
        for (int g = 0; g < 24; ++g) {
            pv_mfma(g);
            if (g < 8 && !(W1_SKIP & 32)) load_k(b, 1, g);
            w1_slice<MASK>(g, stX, sX1, sX2, tile * AK, nk, lane, qvX, huX, luX);
            W1_GAP();
        }
        rescale_o(stX, oX1a, oX2a, oX1b, oX2b);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        W1_GAP();
#pragma unroll
        for (int g = 0; g < 24; ++g) {
            qk_mfma(g);
            if (g < 8) { if (!(W1_SKIP & 32)) load_v(b, 0, g); }
            else if (g < 16 && !(W1_SKIP & 64)) issue_piece(dma_tile, dma_buf, g - 8);
            w1_slice<MASK>(g, stY, sY1, sY2, tile * AK, nk, lane, qvY, huY, luY);
            W1_GAP();
        }
        rescale_o(stY, oY1a, oY2a, oY1b, oY2b);
        W1_GAP();
#pragma unroll
        for (int g = 0; g < 24; ++g) {
            pv_mfma(g);
            if (g < 8 && !(W1_SKIP & 32)) load_k(b ^ 1, 0, g);
            w1_slice<MASK>(g, stX, sX1, sX2, tile * AK + 32, nk, lane, qvX, huX, luX);
            W1_GAP();
        }
        rescale_o(stX, oX1a, oX2a, oX1b, oX2b);
    };
    using T_ = std::true_type; using F_ = std::false_type;
    const bool ragged = (nk & (AK - 1)) != 0;
    if (T == 1) {
        if (ragged) tile_body(T_{}, T_{}, 0); else tile_body(F_{}, T_{}, 0);
    } else {
        tile_body(F_{}, T_{}, 0);
        for (int tile = 1; tile < T - 1; ++tile) tile_body(F_{}, F_{}, tile);
        if (ragged) tile_body(T_{}, F_{}, T - 1); else tile_body(F_{}, F_{}, T - 1);
    }
    {
        const int b = (T - 1) & 1;
#pragma unroll
        for (int g = 0; g < 8; ++g) load_v(b, 1, g);
        if (ragged) {
#pragma unroll
            for (int g = 0; g < 24; ++g) w1_slice<true>(g, stY, sY1, sY2, (T - 1) * AK + 32, nk, lane, qvY, huY, luY);
        } else {
#pragma unroll
            for (int g = 0; g < 24; ++g) w1_slice<false>(g, stY, sY1, sY2, 0, nk, lane, qvY, huY, luY);
        }
        rescale_o(stY, oY1a, oY2a, oY1b, oY2b);
#pragma unroll
        for (int g = 0; g < 24; ++g) pv_mfma(g);
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");
    }
#undef W1_GAP

    auto store = [&](int qrow, float l_run, const f32x16& o1a, const f32x16& o2a, const f32x16& o1b, const f32x16& o2b) {
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        if (qrow >= nq) return;
        const float inv = 1.0f / l_tot;
        const int prow = img * p.Kc + qrow;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                half4 hh, ll;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = half ? (o1b[4 * g4 + e] + o2b[4 * g4 + e] * SPLIT_INV)
                                         : (o1a[4 * g4 + e] + o2a[4 * g4 + e] * SPLIT_INV);
                    _Float16 a, b2;
                    split_f32(v * inv, a, b2, range_flag_of(p.ctrl, img));
                    hh[e] = a; ll[e] = b2;
                }
                const size_t o = panel_index(prow, head * DH + 32 * half + 8 * g4 + 4 * h, p.NIc * p.Kc);
                *reinterpret_cast<half4*>(p.msg.hi + o) = hh;
                *reinterpret_cast<half4*>(p.msg.lo + o) = ll;
            }
        }
    };
    store(qrowX, stX.l_run, oX1a, oX2a, oX1b, oX2b);
    store(qrowY, stY.l_run, oY1a, oY2a, oY1b, oY2b);
}


}  // namespace
