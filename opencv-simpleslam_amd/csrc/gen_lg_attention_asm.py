#!/usr/bin/env python3
"""Generator of the hand-scheduled LightGlue attention kernel for gfx950 (csrc/lg_attention_asm.s).

Same arithmetic, LDS images, DMA schedule and results (bit for bit) as lg_attention_p_kernel in
csrc/lightglue_kernels.hip (and as the compiler-scheduled half-step form of it, scripts/ubench/attn_hs_reference.hpp,
whose schedule this one writes out by hand) - flash-style split-precision attention, 4 waves x 32 queries, S^T = K Q^T so a
lane owns a query column, P fed to P.V straight from registers - but written as ONE instruction stream per
wave in which every 32-cycle MFMA slot carries its share of the softmax arithmetic, the LDS fragment reads
of the next half and the tile DMA.  hipcc cannot produce this stream: with the ~250 registers the kernel
needs it sinks every fragment read to just in front of its MFMA and clumps the softmax into runs with no
matrix instruction beside them (profiles/r03_attention_experiments.md); here registers are assigned by
hand (156 arch VGPRs: O, logits, P, per-lane scalars, addresses; 96 AGPRs: Q, K and V^T fragments), the
element-wise steps that come in (even, odd) pairs use the packed fp32 instructions, and the order is the
schedule.

A 32-key sub-step j of a wave is two halves of 12 MFMAs:
    ODD(j)   O += V^T(j-1) P(j-1) | K fragments of sub-step j+1 | softmax(j): combine, row max, rescale vote,
                                                                    reference, exp2 of elements 0-7
    EVEN(j)  S(j+1) = K(j+1) Q^T  | V^T fragments of sub-step j | exp2 of elements 8-15, hi / lo split -> P(j),
                                                                    row sum; (rare) O *= alpha
One workgroup barrier per 64-key tile, between ODD and EVEN of its first sub-step (as in the hs kernel).

Software-visible hazards of the matrix pipe (no interlock): a VALU read of an MFMA result needs >= 11 issue
slots after the MFMA (8-pass); every such read here sits >= 13 instructions behind it (part 1 starts in slot
3 of the following half; the O rescale runs a whole half later; the epilogue pads with s_nop).

usage: gen_lg_attention_asm.py > lg_attention_asm.s   (build.py does this, assembles it for gfx950 and embeds the code object)
"""
import os
import sys

# timing ablations (results are garbage): ABL = comma list of novalu, nods, nomfma, nodma
ABL = set(filter(None, os.environ.get("ATTN_ASM_ABL", "").split(",")))
# packed fp32 (v_pk_fma_f32 / v_pk_add_f32) for the pairwise element-wise steps: measured SLOWER by 55 us per 8-pair launch
# than two plain instructions each (they do not issue under a running MFMA), kept as a switch for the record
PK = os.environ.get("ATTN_ASM_PK", "0") == "1"
# schedule knobs (experiments; the defaults are the measured best)
NEXP_ODD = int(os.environ.get("ATTN_ASM_NEXP_ODD", "4"))     # exp2 pairs (of 8) taken in the ODD half, the rest in EVEN
DEFER = int(os.environ.get("ATTN_ASM_DEFER", "0"))           # 1: row sums of the EVEN half's pairs + the l update run in slots 0-2 of the next ODD half
DMA_FIRST = int(os.environ.get("ATTN_ASM_DMA_FIRST", "1"))   # first MFMA slot that carries a DMA piece (experiments: 4 = behind the fragment reads)
WEIGHTED = int(os.environ.get("ATTN_ASM_WEIGHTED", "0"))     # 1: spread the VALU stream by issue cost (transcendental 2, else 1) instead of by count
# diagnostic build (never the product): s_memtime stamps around the three kinds of waits of the loop; every wave writes
# {lifetime, tile-barrier wait, fragment wait at ODD, fragment wait at EVEN, stamp-pair overhead, tiles} (cycles) to the
# buffer whose address sits at kernel-argument offset 96
STAMP = os.environ.get("ATTN_ASM_STAMP", "0") == "1"
# r04: P as ONE fp16 plane (the opt-in precision "f16x3p1", sslam_lightglue_set_precision(lg, 2); csrc/gen_lg_attention_asm_p1.py
# generates that kernel): the four V^T.hi x P.lo MFMAs of a sub-step and the two fma_mix per element pair that form P.lo are
# not issued (20 MFMAs per sub-step instead of 24), and the row sum runs over the ROUNDED weights (one v_fma_mix_f32 per
# element from the packed halves) so that o / l stays an exact softmax of slightly perturbed logits - the arithmetic of
# lg_attention_p_kernel with p_single (profiles/r04_split_study.md: zero index flips on every case, token state 2.4e-5).
P1 = os.environ.get("ATTN_ASM_P1", "0") == "1"

out = []
def e(s=""):
    out.append(s)

KERNEL = os.environ.get("ATTN_ASM_KERNEL", "lg_attention_asm_kernel")

# ---------------------------------------------------------------- register map
# arch VGPRs
O1A, O2A, O1B, O2B = 0, 16, 32, 48      # context accumulators (hi.hi + true-scale, cross terms) x two 32-column halves
S1, S2, SV = 64, 80, 96                 # logits hi.hi / cross terms (MFMA D), combined logits -> P values (fp32)
PH, PL = 112, 120                       # P fragments: ph[0] 112-115, ph[1] 116-119, pl[0] 120-123, pl[1] 124-127
V_M, V_L = 128, 129
V_MB = 130                              # even: broadcast source of a packed op
V_TMAX = 131
V_ALPHA = 132                           # even
V_T0 = 133
V_PS0, V_PS1 = 134, 135                 # even pair
V_T1 = 136
V_ROW, V_NINF = 137, 138                # 4 h (key row offset of accumulator register 0), -inf
V_ADDR = 139                            # 139-142: fragment LDS byte addresses, chunk q = 0..3
V_DMA0, V_DMA1 = 143, 144               # DMA lane offsets for even / odd 8-row groups
V_TMP = 145                             # 145-155 scratch (prologue / epilogue)
V_THR = 155                             # m_run + 1.5 (loop only; the prologue uses the scratch up to 154)
ACC_OFF = 156
# AGPRs
A_Q = 0                                 # qh[s] = a[4s..], ql[s] = a[16+4s..]
A_K = 32                                # kh[s] = a[32+4s..], kl[s] = a[48+4s..]
A_V = 64                                # vh[s2i][db] = a[64 + 8 s2i + 4 db ..], vl = a[80 + ...]
N_AGPR = 96

def v(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"
def a(i, n=1):
    return f"a{i}" if n == 1 else f"a[{i}:{i + n - 1}]"

# SGPRs: s[0:1] kernarg, s2 / s3 workgroup id x / y
# s[4:19] eight pointers: Q.hi Q.lo K.hi K.lo VT.hi VT.lo msg.hi msg.lo ; s[20:21] ctrl (of the pair, after set-up)
S_CROSS, S_KC, S_NIC, S_NQB, S_NSLAB, S_MAGIC = "s24", "s25", "s26", "s27", "s28", "s29"
S_IMG, S_HEAD, S_KIMG, S_NQ, S_NK, S_Q0, S_T, S_WAVE, S_TILE = "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39"
S_RSRC = "s[40:43]"
S_DOFF, S_LDS0, S_LDS1, S_DDELTA = "s44", "s45", "s46", "s47"     # DMA: tile byte offset, LDS bases for body 0 / 1, tile delta
S_QV = "s[48:49]"                 # lanes that own a real query
S_M0, S_M1 = "s[50:51]", "s[52:53]"
S_RESC = "s54"                    # rescale flag of the current sub-step
S_TMP, S_TMP2, S_TMP3 = "s55", "s56", "s57"
S_RET, S_SUB, S_SUB2 = "s[60:61]", "s[62:63]", "s[76:77]"
S_RAGGED, S_NKREM = "s64", "s65"
S_NIH, S_LKS, S_Z = "s22", "s23", "s30"      # key split: (image, head) slabs per key range, log2(ranges), this workgroup's range
S_INV2 = "s[68:69]"               # {2^-11, 2^-11}
S_C4E6, S_TLAST, S_CMIN = "s70", "s71", "s72"   # 4.0e6f ; T - 1 ; 2^-14
S_NSCALE, S_C65520, S_CINF = "s73", "s74", "s75"
N_SGPR = 96 if STAMP else 80
KARG = 136 if STAMP else 128      # kernel-argument bytes: 9 pointers, 8 ints, the 3 partial buffers (, the stamp buffer)
S_DBG, S_ACC_BAR, S_ACC_ODD, S_ACC_EVEN, S_T0, S_TA, S_TB, S_ACC_CAL = "s[80:81]", 82, 84, 86, "s[88:89]", 90, 92, 94

def stamped(wait_lines, acc):
    """the wait, bracketed by two s_memtime reads whose difference is added to the 64-bit SGPR counter `acc`"""
    if not STAMP:
        return list(wait_lines)
    return ([f"s_memtime s[{S_TA}:{S_TA + 1}]"] + list(wait_lines) +
            [f"s_memtime s[{S_TB}:{S_TB + 1}]", "s_waitcnt lgkmcnt(0)",
             f"s_sub_u32 s{S_TB}, s{S_TB}, s{S_TA}", f"s_subb_u32 s{S_TB + 1}, s{S_TB + 1}, s{S_TA + 1}",
             f"s_add_u32 s{acc}, s{acc}, s{S_TB}", f"s_addc_u32 s{acc + 1}, s{acc + 1}, s{S_TB + 1}"])

MFMA = "v_mfma_f32_32x32x16_f16"

# ---------------------------------------------------------------- building blocks
def kfrag_off(sub, plane, buf):
    return plane * 16384 + buf * 8192 + sub * 4096
def vfrag_off(db, plane, buf):
    return 32768 + plane * 16384 + buf * 8192 + db * 4096

def k_reads(sub, buf):
    r = []
    for s in range(4):
        r.append(f"ds_read_b128 {a(A_K + 4 * s, 4)}, {v(V_ADDR + s)} offset:{kfrag_off(sub, 0, buf)}")
        r.append(f"ds_read_b128 {a(A_K + 16 + 4 * s, 4)}, {v(V_ADDR + s)} offset:{kfrag_off(sub, 1, buf)}")
    return r
def v_reads(sub, buf):
    r = []
    for s2i in range(2):
        for db in range(2):
            q = 2 * sub + s2i
            r.append(f"ds_read_b128 {a(A_V + 8 * s2i + 4 * db, 4)}, {v(V_ADDR + q)} offset:{vfrag_off(db, 0, buf)}")
            r.append(f"ds_read_b128 {a(A_V + 16 + 8 * s2i + 4 * db, 4)}, {v(V_ADDR + q)} offset:{vfrag_off(db, 1, buf)}")
    return r

def mfma_pv():
    m = []
    for s2i in range(2):
        vh0, vh1 = a(A_V + 8 * s2i, 4), a(A_V + 8 * s2i + 4, 4)
        vl0, vl1 = a(A_V + 16 + 8 * s2i, 4), a(A_V + 16 + 8 * s2i + 4, 4)
        ph, pl = v(PH + 4 * s2i, 4), v(PL + 4 * s2i, 4)
        m += [f"{MFMA} {v(O1A, 16)}, {vh0}, {ph}, {v(O1A, 16)}",
              f"{MFMA} {v(O1B, 16)}, {vh1}, {ph}, {v(O1B, 16)}",
              f"{MFMA} {v(O2A, 16)}, {vl0}, {ph}, {v(O2A, 16)}",
              f"{MFMA} {v(O2B, 16)}, {vl1}, {ph}, {v(O2B, 16)}"]
        if not P1:
            m += [f"{MFMA} {v(O1A, 16)}, {vh0}, {pl}, {v(O1A, 16)}",
                  f"{MFMA} {v(O1B, 16)}, {vh1}, {pl}, {v(O1B, 16)}"]
    return m

def mfma_qk():
    m = []
    for s in range(4):
        kh, kl = a(A_K + 4 * s, 4), a(A_K + 16 + 4 * s, 4)
        qh, ql = a(A_Q + 4 * s, 4), a(A_Q + 16 + 4 * s, 4)
        c1 = "0" if s == 0 else v(S1, 16)
        c2 = "0" if s == 0 else v(S2, 16)
        if "qk3acc" in ABL:     # timing experiment: the lo.hi terms in a third accumulator (no adjacent dependent MFMAs)
            m += [f"{MFMA} {v(S1, 16)}, {kh}, {qh}, {c1}",
                  f"{MFMA} {v(S2, 16)}, {kh}, {ql}, {c2}",
                  f"{MFMA} {v(SV, 16)}, {kl}, {qh}, {'0' if s == 0 else v(SV, 16)}"]
            continue
        m += [f"{MFMA} {v(S1, 16)}, {kh}, {qh}, {c1}",
              f"{MFMA} {v(S2, 16)}, {kh}, {ql}, {c2}",
              f"{MFMA} {v(S2, 16)}, {kl}, {qh}, {v(S2, 16)}"]
    if "qkchain" in ABL:        # timing experiment: everything in ONE dependent chain
        m = [x.replace(v(S2, 16), v(S1, 16)) for x in m]
    return m

def softmax_part1(mask, tag):
    """combine, (mask), row max, rescale vote, new reference, alpha, exponent bias.  sv in v96-111."""
    c = []
    for r in range(0, 16, 2):                                                                    # sv = s2 * 2^-11 + s1
        if PK:
            c.append(f"v_pk_fma_f32 {v(SV + r, 2)}, {v(S2 + r, 2)}, {S_INV2}, {v(S1 + r, 2)}")
        else:
            c.append(f"v_fma_f32 {v(SV + r)}, {v(S2 + r)}, s68, {v(S1 + r)}")
            c.append(f"v_fma_f32 {v(SV + r + 1)}, {v(S2 + r + 1)}, s68, {v(S1 + r + 1)}")
    if mask:
        for r in range(16):
            rowc = (r & 3) + 8 * (r >> 2)
            c.append(f"s_sub_i32 {S_TMP}, {S_NKREM}, {rowc}")
            c.append(f"v_cmp_le_i32_e32 vcc, {S_TMP}, {v(V_ROW)}")                               # 4 h + rowc >= nk - kbase
            c.append(f"v_cndmask_b32_e32 {v(SV + r)}, {v(SV + r)}, {v(V_NINF)}, vcc")
    c.append(f"v_max3_f32 {v(V_TMAX)}, {v(SV)}, {v(SV + 1)}, {v(SV + 2)}")
    for r in range(3, 15, 2):
        c.append(f"v_max3_f32 {v(V_TMAX)}, {v(V_TMAX)}, {v(SV + r)}, {v(SV + r + 1)}")
    c.append(f"v_max_f32_e32 {v(V_TMAX)}, {v(V_TMAX)}, {v(SV + 15)}")
    # the other 16 keys of this query live in lane ^ 32
    c.append(f"v_mov_b32_e32 {v(V_T0)}, {v(V_TMAX)}")
    c.append("s_nop 1")
    c.append(f"v_permlane32_swap_b32_e32 {v(V_TMAX)}, {v(V_T0)}")
    c.append(f"v_max_f32_e32 {v(V_TMAX)}, {v(V_TMAX)}, {v(V_T0)}")
    # rescale = any(qvalid && tmax > m_run + 1.5): the new reference, alpha and the exponent bias are only computed
    # when some query of the wave asks for it (wave-uniform call); otherwise m_run, mb stay and alpha is 1
    c.append(f"v_cmp_gt_f32_e32 vcc, {v(V_TMAX)}, {v(V_THR)}")
    c.append(f"s_and_b64 {S_M0}, vcc, {S_QV}")
    c.append(f"s_cmp_lg_u64 {S_M0}, 0")
    c.append(f"s_cselect_b32 {S_RESC}, 1, 0")
    c.append(f"s_cbranch_scc0 .Lnonew_{tag}\n    s_swappc_b64 {S_RET}, {S_SUB2}\n.Lnonew_{tag}:")     # one item: never split by an MFMA
    return c

def softmax_exp(pair):
    """exp2 of elements 2 pair, 2 pair + 1 (p stays in the sv registers)."""
    r0, r1 = SV + 2 * pair, SV + 2 * pair + 1
    if PK:
        return [f"v_pk_add_f32 {v(r0, 2)}, {v(r0, 2)}, {v(V_MB, 2)} op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
                f"v_exp_f32_e32 {v(r0)}, {v(r0)}",
                f"v_exp_f32_e32 {v(r1)}, {v(r1)}"]
    return [f"v_sub_f32_e32 {v(r0)}, {v(r0)}, {v(V_MB)}",
            f"v_sub_f32_e32 {v(r1)}, {v(r1)}, {v(V_MB)}",
            f"v_exp_f32_e32 {v(r0)}, {v(r0)}",
            f"v_exp_f32_e32 {v(r1)}, {v(r1)}"]
def softmax_sum(pair):
    """psum0 += p[even], psum1 += p[odd]"""
    r0, r1 = SV + 2 * pair, SV + 2 * pair + 1
    if pair == 0:
        return [f"v_mov_b32_e32 {v(V_PS0)}, {v(r0)}", f"v_mov_b32_e32 {v(V_PS1)}, {v(r1)}"]
    if PK:
        return [f"v_pk_add_f32 {v(V_PS0, 2)}, {v(V_PS0, 2)}, {v(r0, 2)}"]
    return [f"v_add_f32_e32 {v(V_PS0)}, {v(V_PS0)}, {v(r0)}", f"v_add_f32_e32 {v(V_PS1)}, {v(V_PS1)}, {v(r1)}"]

def softmax_sum_p1(pair):
    """P1: psum0 += float(ph[even element]), psum1 += float(ph[odd element]) - the sums of the ROUNDED weights, from the packed halves"""
    hreg = PH + pair
    c0 = "0" if pair == 0 else v(V_PS0)
    c1 = "0" if pair == 0 else v(V_PS1)
    return [f"v_fma_mix_f32 {v(V_PS0)}, {v(hreg)}, 1.0, {c0} op_sel_hi:[1,0,0]",
            f"v_fma_mix_f32 {v(V_PS1)}, {v(hreg)}, 1.0, {c1} op_sel:[1,0,0] op_sel_hi:[1,0,0]"]

def softmax_split(pair):
    """hi = fp16(p) packed, lo = fp16(p - hi) (one fma_mix each): P fragment registers."""
    r0, r1 = SV + 2 * pair, SV + 2 * pair + 1
    hreg, lreg = PH + pair, PL + pair          # element pair r -> word r of (ph[0], ph[1])
    if P1:
        return [f"v_cvt_pk_f16_f32 {v(hreg)}, {v(r0)}, {v(r1)}"]
    return [f"v_cvt_pk_f16_f32 {v(hreg)}, {v(r0)}, {v(r1)}",
            f"v_fma_mixlo_f16 {v(lreg)}, {v(hreg)}, -1.0, {v(r0)} op_sel_hi:[1,0,0]",
            f"v_fma_mixhi_f16 {v(lreg)}, {v(hreg)}, -1.0, {v(r1)} op_sel:[1,0,0] op_sel_hi:[1,0,0]"]

def softmax_lsum():
    c = [f"v_add_f32_e32 {v(V_PS0)}, {v(V_PS0)}, {v(V_PS1)}",
         f"v_fmac_f32_e32 {v(V_PS0)}, {v(V_L)}, {v(V_ALPHA)}",      # ps0 = l_run * alpha + (psum0 + psum1)
         f"v_mov_b32_e32 {v(V_L)}, {v(V_PS0)}"]
    if DEFER:                                                        # (the rescale subroutine ran before this: reset here)
        c.append(f"v_mov_b32_e32 {v(V_ALPHA)}, 1.0")
    return c

def deferred_block():
    """row sums of the pairs whose exp2 ran in the previous EVEN half, then the l update (sub-step j - 1)"""
    c = []
    for p in range(NEXP_ODD, 8):
        c += softmax_sum(p)
    return c + softmax_lsum()

def interleave(*streams):
    """round-robin merge keeping each stream's order (hides the latency of exp / cvt behind the neighbour)."""
    res, its = [], [list(s) for s in streams]
    while any(its):
        for s in its:
            if s:
                res.append(s.pop(0))
    return res

def dma_issue(ldsbase_sgpr):
    """8 LDS-DMA pieces (8 rows x 128 B each) of this wave's plane: tile at byte offset S_DOFF -> LDS `ldsbase_sgpr`."""
    c = []
    for rg in range(8):
        grp = [f"s_add_u32 m0, {ldsbase_sgpr}, {rg * 1024}"]
        grp.append(f"s_add_u32 {S_TMP3}, {S_DOFF}, {rg * 1024}" if rg else "s_nop 0")
        soff = S_TMP3 if rg else S_DOFF
        vo = V_DMA0 if rg % 2 == 0 else V_DMA1
        grp.append(f"buffer_load_dwordx4 {v(vo)}, {S_RSRC}, {soff} offen lds")
        c.append(grp)
    return c

# ---------------------------------------------------------------- half-step scheduler
def _cost(x):
    if not WEIGHTED:
        return 1.0
    if x.startswith("v_exp_f32"):
        return 2.0
    if x.startswith("s_") and not x.startswith("s_nop"):
        return 0.25
    return 1.0

def emit_half(mfmas, ds, valu, dma=None, valu_start=0, comment="", head=None):
    """12 (or 0) MFMAs; one ds_read behind each of the first MFMAs, the VALU stream spread over the slots from
    `valu_start` on (`head`: a second stream for the slots before it), one DMA piece (m0 + offset set-up + load) per
    slot from slot 1 on."""
    e(f"    ; ---- {comment}")
    head = list(head or [])
    if "novalu" in ABL: valu, head = [], []
    if "fakevalu" in ABL:       # the same number of VALU instructions, all independent full-rate fmas
        valu = [f"v_fma_f32 {v(SV + i % 16)}, {v(SV + i % 16)}, {v(V_ROW)}, {v(V_NINF)}" for i, x in enumerate(valu) if x.startswith("v_")]
    if "notrans" in ABL:        # transcendentals replaced by moves
        valu = [x.replace("v_exp_f32_e32", "v_mov_b32_e32") for x in valu]
    if "nopk" in ABL:           # drop the packed instructions
        valu = [x for x in valu if not x.startswith("v_pk_")]
    if "nods" in ABL: ds = []
    if "nodma" in ABL: dma = None
    if "nomfma" in ABL: mfmas = ["s_nop 0"] * len(mfmas)
    if not mfmas:
        for x in ds: e("    " + x)
        for x in head + valu: e("    " + x)
        return
    nslot = len(mfmas)
    per = [[] for _ in range(nslot)]
    def spread(items, lo, hi):
        tot = sum(_cost(x) for x in items) or 1.0
        acc = 0.0
        for x in items:
            per[lo + min(hi - lo - 1, int(acc / tot * (hi - lo)))].append(x)
            acc += _cost(x)
    if head:
        spread(head, 0, valu_start)
    spread(valu, valu_start if head or valu_start else 0, nslot)
    dsl = list(ds)
    dml = list(dma or [])
    for sl in range(nslot):
        e("    " + mfmas[sl])
        if dsl:
            e("    " + dsl.pop(0))
        if dml and sl >= DMA_FIRST:
            for _ in range(2 if len(dml) > nslot - sl else 1):      # (a late first slot: two pieces per slot where needed)
                for x in dml.pop(0):
                    e("    " + x)
        for x in per[sl]:
            e("    " + x)
    assert not dsl and not dml

def rescale_call(tag):
    if ABL & {"novalu", "fakevalu", "notrans", "nopk"}:
        return
    e(f"    s_cmp_eq_u32 {S_RESC}, 0")
    e(f"    s_cbranch_scc1 .Lnoresc_{tag}")
    e(f"    s_swappc_b64 {S_RET}, {S_SUB}")
    e(f".Lnoresc_{tag}:")

def body(b, last, mask, tag):
    """One 64-key tile (two sub-steps) on buffer b."""
    for sub in range(2):
        st = f"{tag}_s{sub}"
        # ---------------- ODD half
        for x in stamped(["s_waitcnt lgkmcnt(0)"], S_ACC_ODD): e("    " + x)
        if mask:
            e(f"    s_lshl_b32 {S_TMP2}, {S_TILE}, 6")
            e(f"    s_sub_i32 {S_NKREM}, {S_NK}, {S_TMP2}")
            if sub:
                e(f"    s_sub_i32 {S_NKREM}, {S_NKREM}, 32")
        if sub == 0:
            kr = k_reads(1, b)
        else:
            kr = [] if last else k_reads(0, b ^ 1)
        ex, sm = [], []
        for p in range(NEXP_ODD):
            ex += softmax_exp(p)
            if not P1:
                sm += softmax_sum(p)
        if DEFER:
            emit_half(mfma_pv(), kr, softmax_part1(mask, st) + ex + sm, valu_start=3, head=deferred_block(),
                      comment=f"ODD  buf {b} sub {sub}: PV(j-1) | K frags(j+1) | sums(j-1), softmax part 1")
        else:
            emit_half(mfma_pv(), kr, ["s_nop 3"] + softmax_part1(mask, st) + ex + sm, valu_start=3,
                      comment=f"ODD  buf {b} sub {sub}: PV(j-1) | K frags(j+1) | softmax part 1")
        dma = None
        if sub == 0:
            # ---------------- tile barrier: own DMA pieces landed; K(tile) and V^T(tile-1) completely read
            for x in stamped(["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"], S_ACC_BAR): e("    " + x)
            if not last:
                # K waves: tile + 2 (clamped) -> buf b ; V^T waves: tile + 1 -> buf b ^ 1 (S_LDSx: the right base per wave)
                e(f"    s_add_u32 {S_TMP2}, {S_TILE}, {S_DDELTA}")
                e(f"    s_min_u32 {S_TMP2}, {S_TMP2}, {S_TLAST}")
                e(f"    s_lshl_b32 {S_DOFF}, {S_TMP2}, 13")
                dma = dma_issue(S_LDS0 if b == 0 else S_LDS1)
        # ---------------- EVEN half
        do_qk = not (last and sub == 1)
        ex, sm, first, rest = [], [], [], []
        for p in range(NEXP_ODD, 8):
            ex += softmax_exp(p)
            sm += softmax_sum(p)
        for p in range(NEXP_ODD):
            first += softmax_split(p)
        for p in range(NEXP_ODD, 8):
            rest += softmax_split(p)
        # exp2 of the remaining elements under the split of the ODD half's, then (sums and) the rest
        if P1:
            assert not DEFER
            sums = []
            for p in range(8):
                sums += softmax_sum_p1(p)
            valu = interleave(ex, first) + rest + sums + softmax_lsum()
        else:
            valu = interleave(ex, first) + ([] if DEFER else sm) + rest + ([] if DEFER else softmax_lsum())
        for x in stamped(["s_waitcnt lgkmcnt(0)"], S_ACC_EVEN): e("    " + x)
        emit_half(mfma_qk() if do_qk else [], v_reads(sub, b), valu, dma=dma, valu_start=0,
                  comment=f"EVEN buf {b} sub {sub}: QK(j+1) | V frags(j) | softmax part 2")
        rescale_call(st)

# ================================================================= the kernel text
e("// lg_attention_asm.s - GENERATED by opencv-simpleslam_amd/csrc/gen_lg_attention_asm.py (do not edit): hand-scheduled split-precision attention")
e("// for gfx950.  See the generator for the design; csrc/lightglue_kernels.hip lg_attention_p_kernel for the arithmetic.")
e("    .text")
e("    .amdgcn_target \"amdgcn-amd-amdhsa--gfx950\"")
e("    .amdhsa_code_object_version 5")
e(f"    .globl {KERNEL}")
e("    .p2align 8")
e(f"    .type {KERNEL},@function")
e(f"{KERNEL}:")
# ---------------------------------------------------------------- prologue
e("    s_load_dwordx16 s[4:19], s[0:1], 0x0")          # 8 pointers
e("    s_load_dwordx2 s[20:21], s[0:1], 0x40")         # ctrl
e("    s_load_dwordx4 s[24:27], s[0:1], 0x48")         # cross, Kc, NIc, nqb
e("    s_load_dwordx2 s[28:29], s[0:1], 0x58")         # nslab, magic = floor(2^32 / nqb) + 1 (nqb > 1)
e("    s_load_dwordx2 s[22:23], s[0:1], 0x60")         # nih = images * heads, log2(key ranges) (0: the kernel normalises and writes the context itself)
if STAMP:
    e(f"    s_load_dwordx2 {S_DBG}, s[0:1], 0x80")
    e(f"    s_memtime {S_T0}")
    for r in (S_ACC_BAR, S_ACC_ODD, S_ACC_EVEN, S_ACC_CAL):
        e(f"    s_mov_b32 s{r}, 0")
        e(f"    s_mov_b32 s{r + 1}, 0")
T = V_TMP
e(f"    v_mov_b32_e32 {v(T)}, v0")                      # workitem id before v0 becomes an accumulator
e("    s_waitcnt lgkmcnt(0)")
# b = wg_y * nqb + wg_x ; XCD-aware remap when nslab % 8 == 0: slab = (b & 7) + 8 * ((b >> 3) / nqb), qb = (b >> 3) % nqb
e(f"    s_mul_i32 {S_TMP}, s3, {S_NQB}")
e(f"    s_add_u32 {S_TMP}, {S_TMP}, s2")                 # b
e(f"    s_and_b32 {S_TMP2}, {S_NSLAB}, 7")
e(f"    s_cmp_lg_u32 {S_TMP2}, 0")
e("    s_cbranch_scc1 .Lnoremap")
e(f"    s_lshr_b32 {S_TMP2}, {S_TMP}, 3")               # idx
e(f"    s_mul_hi_u32 {S_TMP3}, {S_TMP2}, {S_MAGIC}")    # idx / nqb   (idx * nqb < 2^32)
e(f"    s_cmp_eq_u32 {S_NQB}, 1")
e(f"    s_cselect_b32 {S_TMP3}, {S_TMP2}, {S_TMP3}")
e(f"    s_mul_i32 s58, {S_TMP3}, {S_NQB}")
e(f"    s_sub_u32 s59, {S_TMP2}, s58")                   # qb = idx % nqb
e(f"    s_and_b32 {S_TMP}, {S_TMP}, 7")
e(f"    s_lshl_b32 {S_TMP3}, {S_TMP3}, 3")
e(f"    s_add_u32 s58, {S_TMP}, {S_TMP3}")               # slab
e("    s_branch .Lremapped")
e(".Lnoremap:")
e("    s_mov_b32 s58, s3")
e("    s_mov_b32 s59, s2")
e(".Lremapped:")
# slab = z * nih + (img * 4 + head): z = the key range of this workgroup (<= 3)
e(f"    s_mov_b32 {S_Z}, 0")
for _ in range(3):
    e(f"    s_cmp_ge_u32 s58, {S_NIH}")
    e("    s_cbranch_scc0 .Lzdone")
    e(f"    s_sub_u32 s58, s58, {S_NIH}")
    e(f"    s_add_u32 {S_Z}, {S_Z}, 1")
e(".Lzdone:")
e(f"    s_lshr_b32 {S_IMG}, s58, 2")
e(f"    s_and_b32 {S_HEAD}, s58, 3")
e(f"    s_lshl_b32 {S_Q0}, s59, 7")
# ctrl of the pair: 64 bytes each: n[2] at 0, stop at 24, range flag at 40
e(f"    s_lshr_b32 {S_TMP}, {S_IMG}, 1")
e(f"    s_lshl_b32 {S_TMP}, {S_TMP}, 6")
e(f"    s_add_u32 s20, s20, {S_TMP}")
e("    s_addc_u32 s21, s21, 0")
e("    s_load_dwordx2 s[58:59], s[20:21], 0x0")         # n[0], n[1]
e(f"    s_load_dword {S_TMP2}, s[20:21], 0x18")          # stop
e("    s_waitcnt lgkmcnt(0)")
e(f"    s_cmp_lg_u32 {S_TMP2}, 0")
e("    s_cbranch_scc1 .Lend")
e(f"    s_and_b32 {S_TMP}, {S_IMG}, 1")
e(f"    s_cmp_eq_u32 {S_TMP}, 0")
e(f"    s_cselect_b32 {S_NQ}, s58, s59")                 # nq = n[img & 1]
e(f"    s_cselect_b32 {S_NK}, s58, s59")                 # self: the same image
e(f"    s_mov_b32 {S_KIMG}, {S_IMG}")
e(f"    s_cmp_eq_u32 {S_CROSS}, 0")
e("    s_cbranch_scc1 .Lself")
e(f"    s_xor_b32 {S_KIMG}, {S_IMG}, 1")
e(f"    s_cmp_eq_u32 {S_TMP}, 0")
e(f"    s_cselect_b32 {S_NK}, s59, s58")                 # cross: the other image of the pair
e(".Lself:")
e(f"    s_cmp_ge_u32 {S_Q0}, {S_NQ}")
e("    s_cbranch_scc1 .Lend")
e(f"    s_add_u32 {S_T}, {S_NK}, 63")
e(f"    s_lshr_b32 {S_T}, {S_T}, 6")                     # key tiles (>= 1)
e("    s_mov_b32 s68, 0x3a000000")
e("    s_mov_b32 s69, 0x3a000000")
e(f"    s_mov_b32 {S_C4E6}, 0x4a742400")
e(f"    s_mov_b32 {S_CMIN}, 0x38800000")
e(f"    s_mov_b32 {S_NSCALE}, 0xc5000000")
e(f"    s_mov_b32 {S_C65520}, 0x477ff000")
e(f"    s_mov_b32 {S_CINF}, 0x7f800000")
# lane bookkeeping
e(f"    v_lshrrev_b32_e32 {v(T + 1)}, 6, {v(T)}")
e(f"    v_and_b32_e32 {v(T)}, 63, {v(T)}")               # lane
e(f"    v_readfirstlane_b32 {S_WAVE}, {v(T + 1)}")
e(f"    v_lshrrev_b32_e32 {v(T + 2)}, 5, {v(T)}")        # h
e(f"    v_and_b32_e32 {v(T + 3)}, 31, {v(T)}")           # lr
e(f"    v_lshlrev_b32_e32 {v(V_ROW)}, 2, {v(T + 2)}")    # 4 h
e(f"    v_mov_b32_e32 {v(V_NINF)}, 0xff800000")
# qvalid: q0 + 32 wave + lr < nq
e(f"    s_lshl_b32 {S_TMP}, {S_WAVE}, 5")
e(f"    s_add_u32 {S_TMP}, {S_TMP}, {S_Q0}")             # first query of this wave
e(f"    v_add_u32_e32 {v(T + 4)}, {S_TMP}, {v(T + 3)}")  # qrow
e(f"    v_cmp_gt_u32_e64 {S_QV}, {S_NQ}, {v(T + 4)}")
# key tiles [t0, t1) of this workgroup's range: t0 = z T >> lks, t1 = (z + 1) T >> lks (the p kernel's z * ntiles / KS, KS a power of two)
e(f"    s_mul_i32 {S_TMP}, {S_Z}, {S_T}")
e(f"    s_lshr_b32 {S_TILE}, {S_TMP}, {S_LKS}")          # t0: the loop counts absolute tiles
e(f"    s_add_u32 {S_TMP}, {S_TMP}, {S_T}")
e(f"    s_lshr_b32 {S_TMP}, {S_TMP}, {S_LKS}")           # t1
e(f"    s_cmp_ge_u32 {S_TILE}, {S_TMP}")
e("    s_cbranch_scc1 .Lempty")                         # fewer tiles than ranges: o = 0, m = -inf, l = 0
e(f"    s_sub_u32 {S_TLAST}, {S_TMP}, 1")
e(f"    s_and_b32 {S_RAGGED}, {S_NK}, 63")               # != 0: the last tile OF THE IMAGE is masked
e(f"    s_cmp_eq_u32 {S_TMP}, {S_T}")
e(f"    s_cselect_b32 {S_RAGGED}, {S_RAGGED}, 0")
# fragment addresses: addr[q] = lr * 128 + ((2 q + h) ^ ((lr >> 1) & 7)) * 16
e(f"    v_lshrrev_b32_e32 {v(T + 5)}, 1, {v(T + 3)}")
e(f"    v_and_b32_e32 {v(T + 5)}, 7, {v(T + 5)}")        # swz
e(f"    v_lshlrev_b32_e32 {v(T + 6)}, 7, {v(T + 3)}")    # lr * 128
for q in range(4):
    e(f"    v_add_u32_e32 {v(T + 7)}, {2 * q}, {v(T + 2)}")
    e(f"    v_xor_b32_e32 {v(T + 7)}, {v(T + 7)}, {v(T + 5)}")
    e(f"    v_lshl_add_u32 {v(V_ADDR + q)}, {v(T + 7)}, 4, {v(T + 6)}")
# DMA lane offsets: lrow = lane >> 3, lcp = lane & 7: lrow * 128 + (lcp ^ (4 * (rg & 1) + (lrow >> 1))) * 16
e(f"    v_lshrrev_b32_e32 {v(T + 5)}, 3, {v(T)}")        # lrow
e(f"    v_and_b32_e32 {v(T + 6)}, 7, {v(T)}")            # lcp
e(f"    v_lshrrev_b32_e32 {v(T + 7)}, 1, {v(T + 5)}")    # lrow >> 1
e(f"    v_lshlrev_b32_e32 {v(T + 8)}, 7, {v(T + 5)}")    # lrow * 128
e(f"    v_xor_b32_e32 {v(T + 9)}, {v(T + 6)}, {v(T + 7)}")
e(f"    v_lshl_add_u32 {v(V_DMA0)}, {v(T + 9)}, 4, {v(T + 8)}")
e(f"    v_or_b32_e32 {v(T + 7)}, 4, {v(T + 7)}")
e(f"    v_xor_b32_e32 {v(T + 9)}, {v(T + 6)}, {v(T + 7)}")
e(f"    v_lshl_add_u32 {v(V_DMA1)}, {v(T + 9)}, 4, {v(T + 8)}")
# Q fragments: row qi = min(qrow, Kc - 1) of (img, head): byte offset ((img * 4 + head) * Kc + qi) * 128 + 32 s + 16 h
e(f"    s_sub_u32 {S_TMP}, {S_KC}, 1")
e(f"    v_min_u32_e32 {v(T + 4)}, {S_TMP}, {v(T + 4)}")
e(f"    s_lshl_b32 {S_TMP}, {S_IMG}, 2")
e(f"    s_add_u32 {S_TMP}, {S_TMP}, {S_HEAD}")
e(f"    s_mul_i32 {S_TMP}, {S_TMP}, {S_KC}")              # (img * 4 + head) * Kc rows
e(f"    v_add_u32_e32 {v(T + 4)}, {S_TMP}, {v(T + 4)}")
e(f"    v_lshlrev_b32_e32 {v(T + 4)}, 7, {v(T + 4)}")     # * 128 bytes (the planes are far below 4 GiB)
e(f"    v_lshl_add_u32 {v(T + 4)}, {v(T + 2)}, 4, {v(T + 4)}")   # + 16 h bytes
for s in range(4):
    e(f"    global_load_dwordx4 {a(A_Q + 4 * s, 4)}, {v(T + 4)}, s[4:5] offset:{32 * s}")
    e(f"    global_load_dwordx4 {a(A_Q + 16 + 4 * s, 4)}, {v(T + 4)}, s[6:7] offset:{32 * s}")
# DMA resource of this wave's plane: base = plane + (kimg * 4 + head) * Kc * 128 bytes, records = Kc * 128
e(f"    s_lshl_b32 {S_TMP}, {S_KIMG}, 2")
e(f"    s_add_u32 {S_TMP}, {S_TMP}, {S_HEAD}")
e(f"    s_mul_i32 {S_TMP}, {S_TMP}, {S_KC}")
e(f"    s_lshl_b32 {S_TMP2}, {S_TMP}, 7")                  # low 32 bits of the byte offset
e(f"    s_lshr_b32 {S_TMP3}, {S_TMP}, 25")                 # high bits
e(f"    s_cmp_eq_u32 {S_WAVE}, 0")
e("    s_cselect_b32 s40, s8, s10")
e("    s_cselect_b32 s41, s9, s11")
e(f"    s_cmp_lt_u32 {S_WAVE}, 2")
e("    s_cbranch_scc1 .Lkplane")
e(f"    s_cmp_eq_u32 {S_WAVE}, 2")
e("    s_cselect_b32 s40, s12, s14")
e("    s_cselect_b32 s41, s13, s15")
e(".Lkplane:")
e(f"    s_add_u32 s40, s40, {S_TMP2}")
e(f"    s_addc_u32 s41, s41, {S_TMP3}")
e("    s_and_b32 s41, s41, 0xffff")
e(f"    s_lshl_b32 s42, {S_KC}, 7")                        # num_records: the (image, head) slab
e("    s_mov_b32 s43, 0x00020000")
# per-wave DMA constants: K waves fetch tile + 2 into buffer b, V^T waves tile + 1 into buffer b ^ 1
e(f"    s_lshl_b32 {S_TMP}, {S_WAVE}, 14")                 # plane base in LDS
e(f"    s_cmp_lt_u32 {S_WAVE}, 2")
e(f"    s_cselect_b32 {S_DDELTA}, 2, 1")
e(f"    s_cselect_b32 {S_TMP2}, 0, 8192")
e(f"    s_cselect_b32 {S_TMP3}, 8192, 0")                    # (before the adds: they clobber SCC)
e(f"    s_add_u32 {S_LDS0}, {S_TMP}, {S_TMP2}")             # body on buffer 0 fills: K -> buf 0, V^T -> buf 1
e(f"    s_add_u32 {S_LDS1}, {S_TMP}, {S_TMP3}")             # body on buffer 1 fills: K -> buf 1, V^T -> buf 0
# address of the rescale subroutine
e(f"    s_getpc_b64 {S_SUB}")
e(".Lpc_here:")
e("    s_add_u32 s76, s62, .Lnewref_sub-.Lpc_here")
e("    s_addc_u32 s77, s63, 0")
e("    s_add_u32 s62, s62, .Lrescale_sub-.Lpc_here")
e("    s_addc_u32 s63, s63, 0")
# ---- tile t0 of every plane, then K(t0 + 1)
e(f"    s_lshl_b32 {S_DOFF}, {S_TILE}, 13")
e(f"    s_mov_b32 s58, {S_TMP}")                            # plane base, buffer 0
for grp in dma_issue("s58"):
    for ins in grp:
        e("    " + ins)
# state (under the DMA)
for r in range(64):
    e(f"    v_mov_b32_e32 {v(r)}, 0")
for r in list(range(PH, PH + 16)) + ([V_PS0, V_PS1] + list(range(SV, SV + 16)) if DEFER else []):
    e(f"    v_mov_b32_e32 {v(r)}, 0")
e(f"    v_mov_b32_e32 {v(V_M)}, 0xff800000")
e(f"    v_mov_b32_e32 {v(V_THR)}, 0xff800000")
e(f"    v_mov_b32_e32 {v(V_ALPHA)}, 1.0")
e(f"    v_mov_b32_e32 {v(V_MB)}, 0")
e(f"    v_mov_b32_e32 {v(V_L)}, 0")
e("    s_waitcnt vmcnt(0)")
e("    s_barrier")
e(f"    s_cmp_lt_u32 {S_WAVE}, 2")
e("    s_cbranch_scc0 .Lnok1")
e(f"    s_add_u32 {S_TMP2}, {S_TILE}, 1")
e(f"    s_min_u32 {S_TMP2}, {S_TMP2}, {S_TLAST}")
e(f"    s_lshl_b32 {S_DOFF}, {S_TMP2}, 13")
e(f"    s_add_u32 s58, {S_TMP}, 8192")                      # K(1) -> buffer 1
for grp in dma_issue("s58"):
    for ins in grp:
        e("    " + ins)
e(".Lnok1:")
for x in k_reads(0, 0) + v_reads(0, 0):
    e("    " + x)
e("    s_waitcnt lgkmcnt(0)")
for x in mfma_qk():
    e("    " + x)

# ---------------------------------------------------------------- main loop
if STAMP:
    for x in stamped(["s_waitcnt lgkmcnt(0)"], S_ACC_CAL):       # what an empty stamp pair costs
        e("    " + x)
e(".Lloop:")
e(f"    s_cmp_ge_u32 {S_TILE}, {S_TLAST}")
e("    s_cbranch_scc1 .Llast_b0")
body(0, False, False, "b0")
e(f"    s_add_u32 {S_TILE}, {S_TILE}, 1")
e(f"    s_cmp_ge_u32 {S_TILE}, {S_TLAST}")
e("    s_cbranch_scc1 .Llast_b1")
body(1, False, False, "b1")
e(f"    s_add_u32 {S_TILE}, {S_TILE}, 1")
e("    s_branch .Lloop")
for b in range(2):
    e(f".Llast_b{b}:")
    e(f"    s_cmp_lg_u32 {S_RAGGED}, 0")
    e(f"    s_cbranch_scc1 .Llast_b{b}_m")
    body(b, True, False, f"l{b}")
    e("    s_branch .Lfin")
    e(f".Llast_b{b}_m:")
    body(b, True, True, f"l{b}m")
    e("    s_branch .Lfin")

# ---------------------------------------------------------------- the last P.V, normalise, split, store
e(".Lfin:")
if STAMP:
    T_ = V_TMP
    e(f"    s_memtime s[{S_TB}:{S_TB + 1}]")
    e("    s_waitcnt lgkmcnt(0)")
    e(f"    s_sub_u32 s{S_TB}, s{S_TB}, s88")
    e(f"    s_mul_i32 {S_TMP}, s3, {S_NQB}")
    e(f"    s_add_u32 {S_TMP}, {S_TMP}, s2")
    e(f"    s_lshl_b32 {S_TMP}, {S_TMP}, 2")
    e(f"    s_add_u32 {S_TMP}, {S_TMP}, {S_WAVE}")
    e(f"    s_lshl_b32 {S_TMP}, {S_TMP}, 5")                     # 32 bytes per wave
    e(f"    v_mov_b32_e32 {v(T_ + 10)}, {S_TMP}")
    for i, r in enumerate((S_TB, S_ACC_BAR, S_ACC_ODD, S_ACC_EVEN)):
        e(f"    v_mov_b32_e32 {v(T_ + 1 + i)}, s{r}")
    e(f"    v_mov_b32_e32 {v(T_ + 5)}, s{S_ACC_CAL}")
    e(f"    v_mov_b32_e32 {v(T_ + 6)}, {S_T}")
    e(f"    v_mov_b32_e32 {v(T_ + 7)}, 0")
    e(f"    v_mov_b32_e32 {v(T_ + 8)}, 0")
    e("    s_mov_b64 s[90:91], exec")
    e("    s_mov_b64 exec, 1")
    e("    s_nop 1")
    e(f"    global_store_dwordx4 {v(T_ + 10)}, {v(T_ + 1, 4)}, {S_DBG}")
    e(f"    global_store_dwordx4 {v(T_ + 10)}, {v(T_ + 5, 4)}, {S_DBG} offset:16")
    e("    s_mov_b64 exec, s[90:91]")
e("    s_waitcnt lgkmcnt(0)")
for x in mfma_pv():
    e("    " + x)
if DEFER:
    for x in deferred_block():
        e("    " + x)
# l_tot = l(lo half) + l(hi half)
e(f"    v_mov_b32_e32 {v(V_T0)}, {v(V_L)}")
e("    s_nop 1")
e(f"    v_permlane32_swap_b32_e32 {v(V_L)}, {v(V_T0)}")
e(f"    v_add_f32_e32 {v(V_L)}, {v(V_L)}, {v(V_T0)}")
e(f"    s_cmp_lg_u32 {S_LKS}, 0")
e("    s_cbranch_scc1 .Lpart")                          # a key range: the partial (o, m, l) goes to the merge kernel
# inv = 1 / l_tot (IEEE division, as the compiler expands 1.0f / x)
L, D0, D1, D2, D3, D4 = V_L, V_TMAX, V_PS0, V_PS1, V_T1, V_MB
e(f"    v_div_scale_f32 {v(D0)}, {S_M0}, {v(L)}, {v(L)}, 1.0")
e(f"    v_rcp_f32_e32 {v(D1)}, {v(D0)}")
e(f"    v_div_scale_f32 {v(D2)}, vcc, 1.0, {v(L)}, 1.0")
e("    s_nop 3")
e(f"    v_fma_f32 {v(D3)}, -{v(D0)}, {v(D1)}, 1.0")
e(f"    v_fmac_f32_e32 {v(D1)}, {v(D3)}, {v(D1)}")
e(f"    v_mul_f32_e32 {v(D3)}, {v(D2)}, {v(D1)}")
e(f"    v_fma_f32 {v(D4)}, -{v(D0)}, {v(D3)}, {v(D2)}")
e(f"    v_fmac_f32_e32 {v(D3)}, {v(D4)}, {v(D1)}")
e(f"    v_fma_f32 {v(D0)}, -{v(D0)}, {v(D3)}, {v(D2)}")
e(f"    v_div_fmas_f32 {v(D0)}, {v(D0)}, {v(D1)}, {v(D3)}")
e(f"    v_div_fixup_f32 {v(V_ALPHA)}, {v(D0)}, {v(L)}, 1.0")           # inv
# store addresses: plane offset (bytes) = ((head * 2 + half) * R + prow) * 64 + 16 g4 + 8 h, R = NIc * Kc, prow = img * Kc + qrow
e(f"    s_mul_i32 {S_TMP}, {S_NIC}, {S_KC}")                  # R
e(f"    s_mul_i32 {S_TMP2}, {S_IMG}, {S_KC}")
e(f"    s_lshl_b32 {S_TMP3}, {S_WAVE}, 5")
e(f"    s_add_u32 {S_TMP3}, {S_TMP3}, {S_Q0}")
e(f"    s_add_u32 {S_TMP2}, {S_TMP2}, {S_TMP3}")               # img * Kc + first query of the wave
e(f"    v_lshrrev_b32_e32 {v(T + 1)}, 2, {v(V_ROW)}")           # h
e(f"    v_and_b32_e32 {v(T)}, 31, {v(T)}")                      # lr (T still holds the lane)
e(f"    v_add_u32_e32 {v(T)}, {S_TMP2}, {v(T)}")                # prow
e(f"    s_lshl_b32 {S_TMP3}, {S_HEAD}, 1")
e(f"    s_mul_i32 {S_TMP3}, {S_TMP3}, {S_TMP}")                 # head * 2 * R
e(f"    v_add_u32_e32 {v(T)}, {S_TMP3}, {v(T)}")
e(f"    v_lshlrev_b32_e32 {v(T)}, 6, {v(T)}")                   # * 32 halves * 2 bytes
e(f"    v_lshl_add_u32 {v(T)}, {v(T + 1)}, 3, {v(T)}")          # + 4 h halves = 8 h bytes          (half a)
e(f"    s_lshl_b32 {S_TMP}, {S_TMP}, 6")                        # R * 64 bytes: the next 32-column panel
e(f"    v_add_u32_e32 {v(T + 1)}, {S_TMP}, {v(T)}")             #                                   (half b)
e("    s_nop 7")                                              # the last P.V results (>= 11 slots behind the MFMAs)
# value[half][r] = (o1 + o2 * 2^-11) * inv  -> o1 registers in place
for half in range(2):
    for r in range(0, 16, 2):
        o1, o2 = 32 * half + r, 32 * half + 16 + r
        e(f"    v_pk_fma_f32 {v(o1, 2)}, {v(o2, 2)}, {S_INV2}, {v(o1, 2)}")
        e(f"    v_pk_mul_f32 {v(o1, 2)}, {v(o1, 2)}, {v(V_ALPHA, 2)} op_sel_hi:[1,0]")
e(f"    s_and_b64 exec, exec, {S_QV}")                           # only real queries are stored
e("    s_cbranch_execz .Lend")
HI, LO, AMAX = T + 3, T + 5, T + 2           # two dwords each (even-aligned pairs)
e(f"    v_mov_b32_e32 {v(AMAX)}, 0")
for half in range(2):
    base = 32 * half
    addr = T + half
    for g4 in range(4):
        for pr in range(2):
            a0, a1 = base + 4 * g4 + 2 * pr, base + 4 * g4 + 2 * pr + 1
            z0, z1, t0, t1 = T + 7, T + 8, T + 9, T + 10
            e(f"    v_max3_f32 {v(AMAX)}, |{v(a0)}|, |{v(a1)}|, {v(AMAX)}")
            e(f"    v_cmp_lt_f32_e64 {S_M0}, |{v(a0)}|, {S_CMIN}")
            e(f"    v_cmp_lt_f32_e64 {S_M1}, |{v(a1)}|, {S_CMIN}")
            e(f"    v_cndmask_b32_e64 {v(z0)}, {v(a0)}, 0, {S_M0}")
            e(f"    v_cndmask_b32_e64 {v(z1)}, {v(a1)}, 0, {S_M1}")
            e(f"    v_mul_f32_e32 {v(t0)}, 0x45000000, {v(a0)}")
            e(f"    v_mul_f32_e32 {v(t1)}, 0x45000000, {v(a1)}")
            e(f"    v_cvt_pk_f16_f32 {v(HI + pr)}, {v(z0)}, {v(z1)}")
            e("    s_nop 0")
            e(f"    v_fma_mixlo_f16 {v(LO + pr)}, {v(HI + pr)}, {S_NSCALE}, {v(t0)} op_sel_hi:[1,0,0]")
            e(f"    v_fma_mixhi_f16 {v(LO + pr)}, {v(HI + pr)}, {S_NSCALE}, {v(t1)} op_sel:[1,0,0] op_sel_hi:[1,0,0]")
        e("    s_nop 0")
        e(f"    global_store_dwordx2 {v(addr)}, {v(HI, 2)}, s[16:17] offset:{16 * g4}")
        e(f"    global_store_dwordx2 {v(addr)}, {v(LO, 2)}, s[18:19] offset:{16 * g4}")
        e("    s_nop 0")
# a finite |value| >= 65520 left through an fp16 plane: raise the pair's range flag (split_range_check)
e(f"    v_cmp_ge_f32_e64 {S_M0}, {v(AMAX)}, {S_C65520}")
e(f"    v_cmp_lt_f32_e64 {S_M1}, {v(AMAX)}, {S_CINF}")
e(f"    s_and_b64 {S_M0}, {S_M0}, {S_M1}")
e(f"    s_and_b64 exec, exec, {S_M0}")
e("    s_cbranch_execz .Lend")
e(f"    v_mov_b32_e32 {v(T + 7)}, 0")
e(f"    v_mov_b32_e32 {v(T + 8)}, 1")
e(f"    global_store_dword {v(T + 7)}, {v(T + 8)}, s[20:21] offset:40")
e(".Lend:")
e("    s_endpgm")
# ---------------------------------------------------------------- key split: o (unnormalised), m, l of this range -> partial buffers
e(".Lempty:")
for r in list(range(16)) + list(range(32, 48)):
    e(f"    v_mov_b32_e32 {v(r)}, 0")
e(f"    v_mov_b32_e32 {v(V_M)}, 0xff800000")
e(f"    v_mov_b32_e32 {v(V_L)}, 0")
e("    s_branch .Lpart_store")
e(".Lpart:")
e("    s_nop 7")                                              # the last P.V results
for half in range(2):
    for r in range(0, 16, 2):
        o1, o2 = 32 * half + r, 32 * half + 16 + r
        e(f"    v_pk_fma_f32 {v(o1, 2)}, {v(o2, 2)}, {S_INV2}, {v(o1, 2)}")
e(".Lpart_store:")
e("    s_load_dwordx4 s[4:7], s[0:1], 0x68")                  # o_part, m_part   (the Q pointers are dead)
e("    s_load_dwordx2 s[8:9], s[0:1], 0x78")                  # l_part
# pbase = ((z * NIc + img) * 4 + head) * Kc + q0 + 32 wave + lr ; o at pbase * 64 floats (+ 8 g4 + 4 h, +32 for half b)
e(f"    s_mul_i32 {S_TMP}, {S_Z}, {S_NIC}")
e(f"    s_add_u32 {S_TMP}, {S_TMP}, {S_IMG}")
e(f"    s_lshl_b32 {S_TMP}, {S_TMP}, 2")
e(f"    s_add_u32 {S_TMP}, {S_TMP}, {S_HEAD}")
e(f"    s_mul_i32 {S_TMP}, {S_TMP}, {S_KC}")
e(f"    s_lshl_b32 {S_TMP3}, {S_WAVE}, 5")
e(f"    s_add_u32 {S_TMP3}, {S_TMP3}, {S_Q0}")
e(f"    s_add_u32 {S_TMP}, {S_TMP}, {S_TMP3}")
e(f"    v_and_b32_e32 {v(T)}, 31, {v(T)}")                      # lr (T still holds the lane)
e(f"    v_add_u32_e32 {v(T)}, {S_TMP}, {v(T)}")                 # pbase
e(f"    v_lshrrev_b32_e32 {v(T + 1)}, 2, {v(V_ROW)}")           # h
e(f"    v_lshlrev_b32_e32 {v(T + 2)}, 8, {v(T)}")               # * 64 floats (the partial buffers are far below 4 GiB)
e(f"    v_lshl_add_u32 {v(T + 2)}, {v(T + 1)}, 4, {v(T + 2)}")  # + 4 h floats
e(f"    v_lshlrev_b32_e32 {v(T + 3)}, 2, {v(T)}")               # m / l: one float per (range, image, head, query)
e(f"    s_and_b64 exec, exec, {S_QV}")                           # only real queries are stored
e("    s_cbranch_execz .Lend")
e("    s_waitcnt lgkmcnt(0)")
for g4 in range(4):
    e(f"    global_store_dwordx4 {v(T + 2)}, {v(4 * g4, 4)}, s[4:5] offset:{32 * g4}")
    e(f"    global_store_dwordx4 {v(T + 2)}, {v(32 + 4 * g4, 4)}, s[4:5] offset:{128 + 32 * g4}")
e(f"    v_cmp_eq_u32_e32 vcc, 0, {v(T + 1)}")
e("    s_and_b64 exec, exec, vcc")
e("    s_cbranch_execz .Lend")
e(f"    global_store_dword {v(T + 3)}, {v(V_M)}, s[6:7]")
e(f"    global_store_dword {v(T + 3)}, {v(V_L)}, s[8:9]")
e("    s_endpgm")
# ---------------------------------------------------------------- O *= alpha (rare)
e(".Lrescale_sub:")
for r in range(0, 64, 2):
    e(f"    v_pk_mul_f32 {v(r, 2)}, {v(r, 2)}, {v(V_ALPHA, 2)} op_sel_hi:[1,0]")
if not DEFER:
    e(f"    v_mov_b32_e32 {v(V_ALPHA)}, 1.0")                   # the next sub-steps keep the reference
e(f"    s_setpc_b64 {S_RET}")
# ---------------------------------------------------------------- new reference (rare): m_run, alpha, threshold, exponent bias
e(".Lnewref_sub:")
e(f"    v_max_f32_e32 {v(V_T0)}, {v(V_M)}, {v(V_TMAX)}")             # m_new
e(f"    v_sub_f32_e32 {v(V_T1)}, {v(V_M)}, {v(V_T0)}")
e(f"    v_exp_f32_e32 {v(V_ALPHA)}, {v(V_T1)}")                      # alpha = exp2(m_run - m_new)
e(f"    v_mov_b32_e32 {v(V_M)}, {v(V_T0)}")
e(f"    v_add_f32_e32 {v(V_THR)}, 0x3fc00000, {v(V_T0)}")            # m_run + 1.5
e(f"    v_add_f32_e32 {v(V_T1)}, 0xc1600000, {v(V_T0)}")             # mb = |m_run| < 4e6 ? m_run - 14 : m_run
e(f"    v_cmp_lt_f32_e64 {S_M1}, |{v(V_T0)}|, {S_C4E6}")
e(f"    v_cndmask_b32_e64 {v(V_MB)}, {v(V_T0)}, {v(V_T1)}, {S_M1}")
e(f"    s_setpc_b64 {S_RET}")
e(".Lfunc_end:")
e(f"    .size {KERNEL}, .Lfunc_end-{KERNEL}")
# ---------------------------------------------------------------- descriptor + metadata
e("    .rodata")
e("    .p2align 6")
e(f"    .amdhsa_kernel {KERNEL}")
e("        .amdhsa_group_segment_fixed_size 65536")
e("        .amdhsa_private_segment_fixed_size 0")
e(f"        .amdhsa_kernarg_size {KARG}")
e("        .amdhsa_user_sgpr_count 2")
e("        .amdhsa_user_sgpr_kernarg_segment_ptr 1")
e("        .amdhsa_system_sgpr_workgroup_id_x 1")
e("        .amdhsa_system_sgpr_workgroup_id_y 1")
e("        .amdhsa_system_sgpr_workgroup_id_z 0")
e("        .amdhsa_system_vgpr_workitem_id 0")
e(f"        .amdhsa_next_free_vgpr {ACC_OFF + N_AGPR}")
e(f"        .amdhsa_next_free_sgpr {N_SGPR}")
e(f"        .amdhsa_accum_offset {ACC_OFF}")
e("        .amdhsa_reserve_vcc 1")
e("        .amdhsa_float_round_mode_32 0")
e("        .amdhsa_float_round_mode_16_64 0")
e("        .amdhsa_float_denorm_mode_32 3")
e("        .amdhsa_float_denorm_mode_16_64 3")
e("        .amdhsa_dx10_clamp 1")
e("        .amdhsa_ieee_mode 1")
e("    .end_amdhsa_kernel")
e("    .amdgpu_metadata")
e("---")
e("amdhsa.version: [ 1, 2 ]")
e("amdhsa.kernels:")
e(f"  - .name: {KERNEL}")
e(f"    .symbol: {KERNEL}.kd")
e(f"    .kernarg_segment_size: {KARG}")
e("    .group_segment_fixed_size: 65536")
e("    .private_segment_fixed_size: 0")
e("    .kernarg_segment_align: 8")
e("    .wavefront_size: 64")
e(f"    .sgpr_count: {N_SGPR + 6}")
e(f"    .vgpr_count: {ACC_OFF + N_AGPR}")
e(f"    .agpr_count: {N_AGPR}")
e("    .max_flat_workgroup_size: 256")
e("    .args:")
off = 0
for i in range(9):
    e(f"      - {{.size: 8, .offset: {off}, .value_kind: global_buffer, .address_space: global}}")
    off += 8
for i in range(8):
    e(f"      - {{.size: 4, .offset: {off}, .value_kind: by_value}}")
    off += 4
for i in range(4 if STAMP else 3):
    e(f"      - {{.size: 8, .offset: {off}, .value_kind: global_buffer, .address_space: global}}")
    off += 8
assert off == KARG
e("...")
e("    .end_amdgpu_metadata")

sys.stdout.write("\n".join(out) + "\n")
