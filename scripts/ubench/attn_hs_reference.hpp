// attn_hs_reference.hpp - the compiler-scheduled HALF-STEP form of the split-precision attention kernel (r03): the C++
// counterpart of the schedule csrc/gen_lg_attention_asm.py writes by hand, bit-identical to lg_attention_p_kernel and to
// the assembly kernel.  Lived in the product TU as debug_key_split(-2) until the assembly kernel took over every launch
// (no split and key ranges); kept here for scripts/ubench/attn_bench.hip (A/B and ablations, included after the product TU).
#pragma once

// ---- attention, split precision, HALF-STEP form (batched launches, no key split) -----------------
// Same arithmetic, same LDS images and the same three streams as lg_attention_p_kernel, re-cut so that the
// LDS fragment reads run a HALF sub-step ahead of the MFMAs that consume them.  In the p kernel the
// compiler (237 VGPRs, no room for 16 fragments in flight) sinks every ds_read_b128 to just in front of
// its MFMA: "ds_read; s_waitcnt lgkmcnt(0); mfma; mfma" twelve times per sub-step, each wait a full
// LDS round trip during which this wave's half of the matrix pipe idles (ISA of r02: 12 s_waitcnt per
// 24 MFMAs; PMC: 37 % issuing / 34 % issue-stalled / 29 % waiting).  Here a 32-key sub-step j is two halves
//     half A   MFMA  O += V^T(j-1) P(j-1)    (12, on V^T fragments read during the previous half B)
//              LDS   K fragments of sub-step j+1                      (8 ds_read_b128)
//              VALU  softmax(j), first part: combine the two logit accumulators, row maximum, rescale vote
//     half B   MFMA  S(j+1) = K(j+1) Q^T     (12, on the K fragments read during half A)
//              LDS   V^T fragments of sub-step j                      (8 ds_read_b128)
//              VALU  softmax(j), second part: exp2, row sum, hi / lo split -> P(j) (overwrites P(j-1))
// with a scheduling barrier between the halves, so every fragment has half a sub-step (>= 384 matrix-pipe
// cycles) between its read and its use, and the live sets shrink: ONE P fragment set (P(j) replaces
// P(j-1) once its MFMAs are issued), the combined logits (16 registers) instead of a second accumulator pair.
// One workgroup barrier per 64-key tile, between the halves of its even sub-step: K(t) has been read
// completely (its buffer takes K(t+2)), V^T(t-1) too (V^T(t+1)), and V^T(t) / K(t+1), issued one tile
// earlier, are what the next halves read.
template <bool MASK>
__device__ __forceinline__ void attn_softmax_part1(const f32x16& s1, const f32x16& s2, int kbase, int nk, int lane,
                                                   float (&sv)[16], float& m_run, float& mb, float& alpha, bool& rescale,
                                                   bool qvalid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sv[r] = __builtin_fmaf(s2[r], SPLIT_INV, s1[r]);
    if constexpr (MASK) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (kbase + acc_row(r, lane) >= nk) sv[r] = -INFINITY;
    }
    float tmax = fmaxf(sv[0], sv[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) tmax = fmaxf(fmaxf(tmax, sv[r]), sv[r + 1]);
    {   // the other 16 keys of this query live in lane ^ 32
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    rescale = __any(qvalid && tmax > m_run + RESCALE_THR);       // see attn_softmax_step
    if (rescale) {
        const float m_new = fmaxf(m_run, tmax);
        alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
    } else {
        alpha = 1.0f;
    }
    mb = fabsf(m_run) < 4.0e6f ? m_run - P_BIAS : m_run;
}

__device__ __forceinline__ void attn_softmax_part2(const float (&sv)[16], float mb, float alpha, float& l_run,
                                                   half8 (&ph)[2], half8 (&pl)[2]) {
    float psum0 = 0.0f, psum1 = 0.0f;
    typedef unsigned uint4v __attribute__((ext_vector_type(4)));
    uint4v hu[2], lu[2];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float p0 = __builtin_amdgcn_exp2f(sv[2 * r] - mb);
        const float p1 = __builtin_amdgcn_exp2f(sv[2 * r + 1] - mb);
        psum0 += p0; psum1 += p1;
        const float2v pv2 = {p0, p1};
        const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(pv2, sslam::half2v));
        unsigned l2;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l2) : "v"(h2), "v"(p0));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l2) : "v"(h2), "v"(p1));
        hu[r >> 2][r & 3] = h2;
        lu[r >> 2][r & 3] = l2;
    }
    ph[0] = __builtin_bit_cast(half8, hu[0]); ph[1] = __builtin_bit_cast(half8, hu[1]);
    pl[0] = __builtin_bit_cast(half8, lu[0]); pl[1] = __builtin_bit_cast(half8, lu[1]);
    l_run = l_run * alpha + (psum0 + psum1);
}

#ifndef ATTN_HS_SCHED
#define ATTN_HS_SCHED 1      // 1: one fragment read behind each of the first 8 MFMAs of a half, VALU spread over all 12
#endif
#if ATTN_HS_SCHED == 1
#define ATTN_HS_INTERLEAVE(NV)                                                \
    _Pragma("unroll") for (int ig_ = 0; ig_ < 12; ++ig_) {                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    \
        if (ig_ < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       \
        __builtin_amdgcn_sched_group_barrier(0x002, (NV), 0);                 \
    }
#else
#define ATTN_HS_INTERLEAVE(NV)
#endif

__global__ __launch_bounds__(256, 2) void lg_attention_hs_kernel(AttnArgsH p) {
    __shared__ AttnSmemH sm;
    const int nqb = gridDim.x, nslab = gridDim.y;
    int slab, qb;
    {
        const int b = blockIdx.y * gridDim.x + blockIdx.x;
        if ((nslab & 7) == 0) { const int xcd = b & 7, idx = b >> 3; slab = xcd + 8 * (idx / nqb); qb = idx % nqb; }
        else { slab = blockIdx.y; qb = blockIdx.x; }
    }
    const int ih = slab;
    const int img = ih >> 2, head = ih & 3;
    if (ctrl_of(p.ctrl, img).stop) return;
    const int kimg = p.cross ? (img ^ 1) : img;
    const int nq = n_of(p.ctrl, img), nk = n_of(p.ctrl, kimg);
    const int q0 = qb * AQ;
    if (q0 >= nq) return;
    const int t = threadIdx.x, lane = t & 63, h = lane >> 5, lr = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int T = (nk + AK - 1) / AK;                // key tiles (>= 1: an empty image sets stop)

    const size_t qoff = ((size_t)img * NH + head) * p.Kc * DH;
    const size_t koff = ((size_t)kimg * NH + head) * p.Kc * DH;
    const int qi = min(q0 + wave * 32 + lr, p.Kc - 1);
    const bool qvalid = q0 + wave * 32 + lr < nq;
    half8 qh[4], ql[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qh[s] = *reinterpret_cast<const half8*>(p.Q.hi + qoff + (size_t)qi * DH + 16 * s + 8 * h);
        ql[s] = *reinterpret_cast<const half8*>(p.Q.lo + qoff + (size_t)qi * DH + 16 * s + 8 * h);
    }

    // wave w owns plane w (K hi, K lo, V^T hi, V^T lo): 8 DMA instructions of 8 rows per tile
    const _Float16* gplane = (wave == 0 ? p.K.hi : wave == 1 ? p.K.lo : wave == 2 ? p.VT.hi : p.VT.lo) + koff;
    const bool is_v = wave >= 2;
    const int lrow = lane >> 3, lcp = lane & 7;
    auto issue_tile = [&](int tile, int buf) {
        _Float16* dst = wave == 0 ? sm.k_hi[buf] : wave == 1 ? sm.k_lo[buf] : wave == 2 ? sm.vt_hi[buf] : sm.vt_lo[buf];
#pragma unroll
        for (int rg = 0; rg < 8; ++rg) {
            const int row = rg * 8 + lrow;
            const int c = lcp ^ ((row >> 1) & 7);
            const _Float16* src = is_v ? gplane + ((size_t)tile * DH + row) * AK + c * 8
                                       : gplane + (size_t)min(tile * AK + row, p.Kc - 1) * DH + c * 8;
            glds16(src, dst + rg * 8 * DH);
        }
    };
    int koffs[2][4], voffs[2][2][2];                 // fragment offsets (halves), as in lg_attention_p_kernel
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

    f32x16 o1a, o2a, o1b, o2b, s1, s2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o1a[r] = 0.0f; o2a[r] = 0.0f; o1b[r] = 0.0f; o2b[r] = 0.0f; }
    float m_run = -INFINITY, l_run = 0.0f;
    half8 ph[2], pl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { ph[i][e] = (_Float16)0.0f; pl[i][e] = (_Float16)0.0f; }
    half8 kfh[4], kfl[4], vfh[2][2], vfl[2][2];      // fragments of the next half
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    auto fetch_k = [&](int buf, auto sub_c) {
        constexpr int SUB = decltype(sub_c)::value;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kfh[s] = *reinterpret_cast<const half8*>(&sm.k_hi[buf][koffs[SUB][s]]);
            kfl[s] = *reinterpret_cast<const half8*>(&sm.k_lo[buf][koffs[SUB][s]]);
        }
    };
    auto fetch_v = [&](int buf, auto sub_c) {
        constexpr int SUB = decltype(sub_c)::value;
#pragma unroll
        for (int s2i = 0; s2i < 2; ++s2i)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                vfh[s2i][db] = *reinterpret_cast<const half8*>(&sm.vt_hi[buf][voffs[SUB][s2i][db]]);
                vfl[s2i][db] = *reinterpret_cast<const half8*>(&sm.vt_lo[buf][voffs[SUB][s2i][db]]);
            }
    };
    auto mma_pv = [&]() {            // O += V^T P, pl at true scale with the hi.hi products (see lg_attention_p_kernel `pv`)
#pragma unroll
        for (int s2i = 0; s2i < 2; ++s2i) {
            o1a = mfma16(vfh[s2i][0], ph[s2i], o1a);
            o1b = mfma16(vfh[s2i][1], ph[s2i], o1b);
            o2a = mfma16(vfl[s2i][0], ph[s2i], o2a);
            o2b = mfma16(vfl[s2i][1], ph[s2i], o2b);
            o1a = mfma16(vfh[s2i][0], pl[s2i], o1a);
            o1b = mfma16(vfh[s2i][1], pl[s2i], o1b);
        }
    };
    auto mma_qk = [&]() {            // S^T = K Q^T into (s1: hi.hi, s2: cross terms)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            s1 = mfma16(kfh[s], qh[s], s == 0 ? zero16 : s1);
            s2 = mfma16(kfh[s], ql[s], s == 0 ? zero16 : s2);
            s2 = mfma16(kfl[s], qh[s], s2);
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

    // ---- prologue: K(0), V^T(0) in place, K(1) on its way; S(0); V^T fragments for the (empty) P(-1) product
    issue_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!is_v && T > 1) issue_tile(1, 1);
    fetch_k(0, I0{});
    fetch_v(0, I0{});                                     // finite data for P(-1) = 0
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mma_qk();

    auto tile_body = [&](auto mask_c, int tile) {
        constexpr bool MASK = decltype(mask_c)::value;
        const int b = tile & 1;
        const bool more1 = tile + 1 < T;
        float sv[16], mb, alpha; bool rescale;
        // ================= even sub-step j = 2 tile
        // half A: O += V^T(j-1) P(j-1) | K fragments of sub-step j+1 (tile, sub 1) | softmax(j) part 1
        __builtin_amdgcn_sched_barrier(0);
        fetch_k(b, I1{});
        mma_pv();
        attn_softmax_part1<MASK>(s1, s2, tile * AK, nk, lane, sv, m_run, mb, alpha, rescale, qvalid);
        ATTN_HS_INTERLEAVE(3);
        // K(tile) and V^T(tile-1) have been read by this wave; own DMA pieces of K(tile+1) / V^T(tile) landed
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (is_v) { if (more1) issue_tile(tile + 1, b ^ 1); }
        else      { if (tile + 2 < T) issue_tile(tile + 2, b); }
        // half B: S(j+1) = K Q^T | V^T fragments of sub-step j (tile, sub 0) | softmax(j) part 2 -> P(j)
        fetch_v(b, I0{});
        mma_qk();
        attn_softmax_part2(sv, mb, alpha, l_run, ph, pl);
        ATTN_HS_INTERLEAVE(6);
        if (rescale) { o1a *= alpha; o2a *= alpha; o1b *= alpha; o2b *= alpha; }
        // ================= odd sub-step j = 2 tile + 1
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // half A: O += V^T(j-1) P(j-1) | K fragments of sub-step j+1 (tile+1, sub 0; the last tile re-reads its own) | part 1
        if (more1) fetch_k(b ^ 1, I0{}); else fetch_k(b, I0{});
        mma_pv();
        attn_softmax_part1<MASK>(s1, s2, tile * AK + 32, nk, lane, sv, m_run, mb, alpha, rescale, qvalid);
        ATTN_HS_INTERLEAVE(3);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // half B: S(j+1) | V^T fragments of sub-step j (tile, sub 1) | part 2
        fetch_v(b, I1{});
        mma_qk();
        attn_softmax_part2(sv, mb, alpha, l_run, ph, pl);
        ATTN_HS_INTERLEAVE(6);
        if (rescale) { o1a *= alpha; o2a *= alpha; o1b *= alpha; o2b *= alpha; }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    const bool ragged = (nk & (AK - 1)) != 0;
    const int tfull = ragged ? T - 1 : T;
    for (int tile = 0; tile < tfull; ++tile) tile_body(std::false_type{}, tile);
    if (tfull < T) tile_body(std::true_type{}, T - 1);
    mma_pv();                                             // the last sub-step's P

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const int qrow = q0 + wave * 32 + lr;
    if (qrow < nq) {
        // normalise, split and write the context planes (k-panel layout: 4 consecutive d = 8 bytes per plane)
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
                    _Float16 a, b;
                    split_f32(v * inv, a, b, range_flag_of(p.ctrl, img));
                    hh[e] = a; ll[e] = b;
                }
                const size_t o = panel_index(prow, head * DH + 32 * half + 8 * g4 + 4 * h, p.NIc * p.Kc);
                *reinterpret_cast<half4*>(p.msg.hi + o) = hh;
                *reinterpret_cast<half4*>(p.msg.lo + o) = ll;
            }
        }
    }
}

