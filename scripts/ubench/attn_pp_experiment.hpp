// attn_pp_experiment.hpp - the 8-wave ping-pong attention experiment (r02), kept OUT of the product (r03):
// included by scripts/ubench/attn_bench.hip after the product translation unit (it uses its internals).
// Bit-identical to lg_attention_p_kernel, slower (262 vs 236 us per 8-pair launch): profiles/r02_attention_experiments.md.
#pragma once
#define MF mfma16
#define SOFTMAX_STEP attn_softmax_step
#define ATTN_PIN_USE(ph_, pl_, l_)
#define ATTN_PLO_TRUE 1
#define ATTN_ABL 0
namespace {
constexpr int AQ2 = 256;         // queries per 8-wave workgroup
// ---- attention, split precision, PING-PONG form (batched launches, no key split) -------------
// Why: in lg_attention_p_kernel the two waves that share a SIMD (one from each of two co-resident
// workgroups) run the same code in phase, so their MFMA bursts collide on the one matrix pipe and
// their softmax bursts on the one vector issue port: the launch costs MFMA time PLUS softmax time
// (ablation at 8 pairs: 117 us of MFMA alone + 105 us of everything else = 228 us, nothing
// overlapped).  Here a workgroup has EIGHT waves, two per SIMD, and the pair is kept in
// anti-phase by construction: every 32-key sub-step of a wave is an M segment (24 MFMA on
// fragments already in registers, raised priority, no LDS or VALU work) followed by a V segment
// (softmax + P split of that sub-step, the LDS fragment reads of the next M segment, this wave's
// share of the tile DMA), with a workgroup barrier between segments; waves 4-7 (group B) run one
// segment behind waves 0-3 (group A), so while one wave of a SIMD multiplies its partner does
// vector work in the issue slots the MFMAs leave free.
//   global segment g:   A: M(j) at g = 2j, V(j) at 2j+1      B: M(j) at 2j+1, V(j) at 2j+2
// K / V^T tiles (64 keys) sit in two LDS buffers each, filled by LDS-DMA (wave w: rows 8w..8w+7 of
// the hi and the lo plane).  K(t) is read (prefetch of the fragments of sub-steps 2t, 2t+1) in
// segments 4t-1 .. 4t+2, V^T(t) in 4t+1 .. 4t+4; K(t+2) is issued in V(2t+1) and V^T(t+1) in V(2t)
// - after the last reader of the buffer they replace - and each is awaited (counted vmcnt: the
// younger group stays in flight) at the end of the global segment before its first reader.

__global__ __launch_bounds__(512) void lg_attention_pp_kernel(AttnArgsH p) {
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
    const int grp = wave >> 2;                       // 0: group A, 1: group B (one segment behind)
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

    // this wave's share of a tile: rows 8 wave .. 8 wave + 7 of the hi and the lo plane
    const int drow = wave * 8 + (lane >> 3);
    const int dchunk = ((lane & 7) ^ ((drow >> 1) & 7)) * 8;
    auto issue_k = [&](int tile, int buf) {
        const size_t so = koff + (size_t)min(tile * AK + drow, p.Kc - 1) * DH + dchunk;
        glds16(p.K.hi + so, sm.k_hi[buf] + wave * 8 * DH);
        glds16(p.K.lo + so, sm.k_lo[buf] + wave * 8 * DH);
    };
    auto issue_v = [&](int tile, int buf) {
        const size_t so = koff + ((size_t)tile * DH + drow) * AK + dchunk;
        glds16(p.VT.hi + so, sm.vt_hi[buf] + wave * 8 * AK);
        glds16(p.VT.lo + so, sm.vt_lo[buf] + wave * 8 * AK);
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
    half8 ph[2], pl[2];                              // P of the last V segment
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { ph[i][e] = (_Float16)0.0f; pl[i][e] = (_Float16)0.0f; }
    half8 kfh[4], kfl[4], vfh[2][2], vfl[2][2];      // fragments of the next M segment
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
    // M segment: S = K Q^T of this sub-step, O += V^T P of the previous one - registers only
    auto mseg = [&](bool with_qk) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        // a single wave feeds the matrix pipe here: consecutive MFMAs never share an accumulator
        // (s2 every third instruction, the others every sixth)
        static_assert(ATTN_PLO_TRUE, "the ping-pong form keeps the low plane of P unscaled");
        if (with_qk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int sa = 2 * i, sb = 2 * i + 1;
                s2 = MF(kfh[sa], ql[sa], i == 0 ? zero16 : s2);
                s1 = MF(kfh[sa], qh[sa], i == 0 ? zero16 : s1);
                o1a = MF(vfh[i][0], ph[i], o1a);
                s2 = MF(kfl[sa], qh[sa], s2);
                o1b = MF(vfh[i][1], ph[i], o1b);
                o2a = MF(vfl[i][0], ph[i], o2a);
                s2 = MF(kfh[sb], ql[sb], s2);
                s1 = MF(kfh[sb], qh[sb], s1);
                o1a = MF(vfh[i][0], pl[i], o1a);
                s2 = MF(kfl[sb], qh[sb], s2);
                o1b = MF(vfh[i][1], pl[i], o1b);
                o2b = MF(vfl[i][1], ph[i], o2b);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                o1a = MF(vfh[i][0], ph[i], o1a);
                o1b = MF(vfh[i][1], ph[i], o1b);
                o2a = MF(vfl[i][0], ph[i], o2a);
                o2b = MF(vfl[i][1], ph[i], o2b);
                o1a = MF(vfh[i][0], pl[i], o1a);
                o1b = MF(vfh[i][1], pl[i], o1b);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // segment boundary: optional counted wait for this wave's older DMA pieces, LDS reads drained
    // (their buffer may be refilled by the partner group right after the barrier), barrier
    auto boundary = [&](int dma_wait /* -1 none, 0 all, 2 all but the youngest group */) {
        if (dma_wait == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else if (dma_wait == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto vseg_softmax = [&](auto mask_c, int kbase) {
        constexpr bool MASK = decltype(mask_c)::value;
        float alpha; bool rescale;
        SOFTMAX_STEP<MASK>(s1, s2, kbase, nk, lane, m_run, l_run, ph, pl, alpha, rescale, qvalid, false);
        ATTN_PIN_USE(ph, pl, l_run);
        if (rescale) { o1a *= alpha; o2a *= alpha; o1b *= alpha; o2b *= alpha; }
    };

    // ---- prologue: tile 0 of K and V^T, then K(1) in flight; fragments of M(0)
    issue_k(0, 0);
    issue_v(0, 0);
    boundary(0);
    if (T > 1) issue_k(1, 1);
    fetch_k(0, std::integral_constant<int, 0>{});
    fetch_v(0, std::integral_constant<int, 0>{});        // P(-1) = 0 meets finite data
    if (grp == 1) boundary(-1);                           // group B idles through global segment 0

    auto tile_body = [&](auto mask_c, int tile) {
        const bool more1 = tile + 1 < T, more2 = tile + 2 < T;
        const int b = tile & 1;
        // M(2t)
        mseg(true);
        boundary(grp == 0 ? (more1 ? 2 : 0) : -1);              // A: V^T(t) landed
        // V(2t): V^T(t+1) on its way, fragments of M(2t+1), softmax of sub-step 2t
#if !(ATTN_ABL & 8)
        if (more1) issue_v(tile + 1, b ^ 1);
#endif
#if !(ATTN_ABL & 4)
        fetch_k(b, std::integral_constant<int, 1>{});
        fetch_v(b, std::integral_constant<int, 0>{});
#endif
        vseg_softmax(mask_c, tile * AK);
        boundary(grp == 1 && more1 ? 2 : -1);                    // B: K(t+1) landed
        // M(2t+1)
        mseg(true);
        boundary(grp == 0 && more1 ? 2 : -1);                    // A: K(t+1) landed
        // V(2t+1): K(t+2) on its way, fragments of M(2t+2), softmax of sub-step 2t+1
#if !(ATTN_ABL & 8)
        if (more2) issue_k(tile + 2, b);
#endif
#if !(ATTN_ABL & 4)
        if (more1) fetch_k(b ^ 1, std::integral_constant<int, 0>{});
        fetch_v(b, std::integral_constant<int, 1>{});
#endif
        vseg_softmax(mask_c, tile * AK + 32);
        boundary(grp == 1 && more1 ? (more2 ? 2 : 0) : -1);      // B: V^T(t+1) landed
    };
    const bool ragged = (nk & (AK - 1)) != 0;
    const int tfull = ragged ? T - 1 : T;
    for (int tile = 0; tile < tfull; ++tile) tile_body(std::false_type{}, tile);
    if (tfull < T) tile_body(std::true_type{}, T - 1);
    mseg(false);                                          // the last sub-step's P.V
    if (grp == 0) boundary(-1);                           // group A idles through the last global segment

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

}  // namespace
