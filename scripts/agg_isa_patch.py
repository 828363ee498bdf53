"""Hand-patch the compiler's assembly of al_aggregate_kernel (the failing, packed-fp32 shape) ONE property at a time, for
scripts/agg_victim_run.py LIB:CODE_OBJECT (profiles/r06_aggregate_rnorm_diagnosis.md, section 5).
usage: agg_isa_patch.py IN.s OUT.s MODE
  identity            the assembly as the compiler wrote it (checks the assemble / load / launch path)
  pk_split:SEL        packed fp32 instructions -> two single-width ones (same operands, same rounding; v86 / v87 as temporaries where
                      the destination overlaps a source).  SEL = all | loaded (those with a source register some global_load wrote
                      since the last s_waitcnt-free point, i.e. a register that is a load destination anywhere in the kernel) |
                      notloaded | a-b,c,... (ordinals of the kernel's packed instructions, 0-based)
  pk_keep:SEL         the complement: every packed instruction split EXCEPT the selected ones
  pk_nop:SEL:N        s_nop N directly behind the selected packed instructions
  load_nop:N          s_nop N behind every global_load_dword
  war_nop:N           s_nop N behind every global_load_dword whose NEXT instruction overwrites one of its address registers
  wait_nop:N          s_nop N behind every s_waitcnt vmcnt
  ins_before:K:a;b / ins_after:K:a;b / repl:K:a;b     instructions in front of / behind / instead of packed instruction K
  sub_near:K:OLD=>a;b the instruction OLD nearest to packed instruction K replaced
  probe:REG:K         v88 = REG in front of the K-th packed instruction, stored where 1 / ||F|| was (see apply())
  MODE+MODE           several of the above
  list                print the packed instructions with their ordinals and whether a source is a load destination; no output file
IN.s: `hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAL_AGG_FAST_SELU=2 -DAL_AGG_PACKED=1 --cuda-device-only -S scripts/ubench/agg_victim.hip`."""
import re, sys

KERNEL = "_ZN12_GLOBAL__N_119al_aggregate_kernelENS_3PyrEPKfPfS3_m"
T0, T1 = 86, 87


def regs_of(tok):
    """registers an operand names: ('v', [n, ...]) / ('s', [...]) / ('c', text)"""
    tok = tok.strip()
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", tok)
    if m:
        return m.group(1), list(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.fullmatch(r"([vs])(\d+)", tok)
    if m:
        return m.group(1), [int(m.group(2))]
    return "c", tok


def parse(line):
    body = line.split(";")[0].strip()
    if not body or body.startswith(".") or body.endswith(":"):
        return None
    parts = body.split(None, 1)
    mn = parts[0]
    rest = parts[1] if len(parts) > 1 else ""
    mods = {}
    for m in re.finditer(r"(op_sel_hi|op_sel|neg_lo|neg_hi):\[([\d,]+)\]", rest):
        mods[m.group(1)] = [int(x) for x in m.group(2).split(",")]
    rest = re.sub(r"(op_sel_hi|op_sel|neg_lo|neg_hi):\[[\d,]+\]", "", rest).strip()
    ops = [o.strip() for o in rest.split(",")] if rest else []
    return mn, ops, mods


def half(tok, hi):
    kind, r = regs_of(tok)
    if kind == "c":
        assert not hi, f"high half of a constant: {tok}"
        return tok
    assert len(r) == 2, tok
    return f"{kind}{r[1 if hi else 0]}"


def split_pk(mn, ops, mods):
    """v_pk_{mul,add,fma}_f32 -> single-width instructions with the same value per half"""
    n = len(ops) - 1
    sel = mods.get("op_sel", [0] * n); selh = mods.get("op_sel_hi", [1] * n)
    nlo = mods.get("neg_lo", [0] * n); nhi = mods.get("neg_hi", [0] * n)
    base = {"v_pk_mul_f32": "v_mul_f32_e64", "v_pk_add_f32": "v_add_f32_e64", "v_pk_fma_f32": "v_fma_f32"}[mn]
    _, d = regs_of(ops[0])
    lo_src = [("-" if nlo[i] else "") + half(ops[1 + i], sel[i]) for i in range(n)]
    hi_src = [("-" if nhi[i] else "") + half(ops[1 + i], selh[i]) for i in range(n)]
    hi_reads = {s.lstrip("-") for s in hi_src}
    out = []
    if f"v{d[0]}" in hi_reads:                      # the low result would clobber an operand of the high one
        out.append(f"\t{base} v{T0}, {', '.join(lo_src)}")
        out.append(f"\t{base} v{d[1]}, {', '.join(hi_src)}")
        out.append(f"\tv_mov_b32_e32 v{d[0]}, v{T0}")
    else:
        out.append(f"\t{base} v{d[0]}, {', '.join(lo_src)}")
        out.append(f"\t{base} v{d[1]}, {', '.join(hi_src)}")
    return out


def main():
    src, dst, mode = sys.argv[1], sys.argv[2], sys.argv[3]
    lines = open(src).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    load_dst = set()
    for l in body:
        p = parse(l)
        if p and p[0].startswith("global_load_dword"):
            load_dst.update(regs_of(p[1][0])[1])
    pk = []                                              # (index in body, parsed, reads a load destination)
    for i, l in enumerate(body):
        p = parse(l)
        if p and p[0].startswith("v_pk_") and p[0].endswith("_f32"):
            reads = set()
            for o in p[1][1:]:
                k, r = regs_of(o)
                if k == "v":
                    reads.update(r)
            pk.append((i, p, bool(reads & load_dst)))
    name = "list" if mode == "list" else ""
    if name == "list":
        for k, (i, p, ld) in enumerate(pk):
            print(f"{k:3d} line {i:4d} {'LOADED' if ld else '      '} {body[i].strip()}")
        print(f"{len(pk)} packed instructions, {sum(1 for x in pk if x[2])} read a register that is a load destination somewhere in the kernel")
        return
    out = list(body)
    for one in mode.split("+"):
        lines = apply(one, body, out, pk, lines, start, end)
    open(dst, "w").write("\n".join(lines[:start] + out + lines[end + 1:]))


def apply(mode, body, out, pk, lines, start, end):
    name, _, arg = mode.partition(":")
    if name in ("pk_split", "pk_keep"):
        if arg == "all":
            chosen = set(range(len(pk)))
        elif arg == "loaded":
            chosen = {k for k, x in enumerate(pk) if x[2]}
        elif arg == "notloaded":
            chosen = {k for k, x in enumerate(pk) if not x[2]}
        else:
            chosen = set()
            for piece in arg.split(","):
                a, _, b = piece.partition("-")
                chosen.update(range(int(a), int(b or a) + 1))
        if name == "pk_keep":                            # everything BUT the named ones is split
            chosen = set(range(len(pk))) - chosen
        for k in chosen:
            i, p, _ = pk[k]
            out[i] = out[i].replace(body[i], "\n".join(split_pk(*p)))
        print(f"pk_split: {len(chosen)} of {len(pk)} packed instructions split")
    elif name == "pk_nop":                                # pk_nop:SEL:N - s_nop N directly behind the selected packed instructions
        sel, _, n = arg.partition(":")
        chosen = set()
        for piece in sel.split(","):
            a, _, b = piece.partition("-")
            chosen.update(range(int(a), int(b or a) + 1))
        for k in chosen:
            out[pk[k][0]] = out[pk[k][0]] + f"\n\ts_nop {int(n)}"
        print(f"pk_nop: s_nop {n} behind {len(chosen)} packed instructions")
    elif name in ("load_nop", "war_nop", "wait_nop"):
        n = int(arg); cnt = 0
        parsed = [parse(l) for l in body]
        for i, p in enumerate(parsed):
            if not p:
                continue
            hit = False
            if name == "wait_nop":
                hit = p[0] == "s_waitcnt" and "vmcnt" in body[i]
            elif p[0].startswith("global_load_dword"):
                if name == "load_nop":
                    hit = True
                else:
                    addr = set(regs_of(p[1][1])[1])
                    nxt = next((q for q in parsed[i + 1:] if q), None)
                    if nxt and nxt[0].startswith("v_") and nxt[1]:
                        k, r = regs_of(nxt[1][0])
                        hit = k == "v" and bool(set(r) & addr)
            if hit:
                out[i] = out[i] + f"\n\ts_nop {n}"; cnt += 1
        print(f"{name}: s_nop {n} in {cnt} places")
    elif name in ("ins_before", "ins_after", "repl"):      # ins_before:K:asm;asm - text in front of / behind / instead of packed instruction K
        k, _, text = arg.partition(":")
        i = pk[int(k)][0]
        text = "\n".join("\t" + t.strip() for t in text.split(";"))
        out[i] = text + "\n" + out[i] if name == "ins_before" else out[i] + "\n" + text if name == "ins_after" else text
        print(f"{name}: packed instruction {k}")
    elif name == "sub_near":                               # sub_near:K:OLD=>NEW;NEW - the instruction OLD nearest to packed instruction K replaced
        k, _, rest = arg.partition(":")
        old, _, new = rest.partition("=>")
        i = pk[int(k)][0]
        cand = sorted((abs(j - i), j) for j in range(max(0, i - 12), min(len(body), i + 12)) if body[j].strip() == old.strip())
        assert cand, f"{old!r} not near packed instruction {k}"
        out[cand[0][1]] = "\n".join("\t" + t.strip() for t in new.split(";"))
        print(f"sub_near: line {cand[0][1]} ({cand[0][1] - i:+d} from packed instruction {k})")
    elif name == "probe":
        # probe:REG:K - v88 (a register added to the kernel's allocation) = REG just in front of the K-th packed instruction; the kernel
        # then stores v88 where it stored 1 / ||F||, so the comparison of scripts/agg_victim_run.py shows that register's value
        reg, _, k = arg.partition(":")
        i = pk[int(k)][0]
        out[i] = f"\tv_mov_b32_e32 v88, {reg}\n" + out[i]
        st = [j for j, l in enumerate(body) if l.strip() == "global_store_dword v[2:3], v1, off"]
        assert len(st) == 1, st
        out[st[0]] = "\tglobal_store_dword v[2:3], v88, off"
        kd = next(j for j, l in enumerate(lines) if l.strip().startswith(".amdhsa_kernel " + KERNEL))
        for j in range(kd, kd + 60):
            if ".amdhsa_next_free_vgpr" in lines[j]:
                lines[j] = "\t\t.amdhsa_next_free_vgpr 105"
            if ".amdhsa_accum_offset" in lines[j]:
                lines[j] = "\t\t.amdhsa_accum_offset 96"
        md = next(j for j, l in enumerate(lines) if l.strip() == ".name:           " + KERNEL)
        for j in range(md, md + 14):
            if ".vgpr_count:" in lines[j]:
                lines[j] = "    .vgpr_count:     96"
        print(f"probe: v88 = {reg} in front of packed instruction {k}, stored in place of 1/||F||")
    elif name != "identity":
        raise SystemExit(f"unknown mode {mode}")
    return lines


main()
