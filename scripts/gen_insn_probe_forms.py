"""Every VALU instruction FORM of the built product (mnemonic + modifiers + operand kinds and widths), rewritten onto the fixed
registers of scripts/ubench/insn_probe.hip: sources v[10:13], v[14:17], v[18:21], destination v[30:33], SGPR operands -> VGPRs of the
same width, carry / condition operands -> vcc.  One exemplar per form is taken from `llvm-objdump -d` of the library's code objects.
Writes scripts/ubench/insn_probe_forms.inc (an X-macro list) and prints what it left out and why.
usage: gen_insn_probe_forms.py            (run after the product is built; needs /opt/rocm/lib/llvm/bin)
Left out by design: MFMA and AccVGPR moves (the aggressors themselves / no VALU datapath), v_readlane / v_readfirstlane / v_writelane
(scalar side), v_cmpx (writes EXEC), v_nop."""
import importlib.util, re, subprocess, sys, tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
spec = importlib.util.spec_from_file_location("g", ROOT / "opencv-simpleslam_amd" / "isa_guard.py")
guard = importlib.util.module_from_spec(spec); spec.loader.exec_module(guard)
LLVM = Path("/opt/rocm/lib/llvm/bin")

SKIP = re.compile(r"v_mfma|v_smfmac|v_accvgpr|v_readlane|v_readfirstlane|v_writelane|v_cmpx|v_nop|v_swap|v_permlane(?!32_swap|16_swap)")
MOD = re.compile(r"\s+(op_sel|op_sel_hi|neg_lo|neg_hi|clamp|mul:|div:|dst_sel|dst_unused|src0_sel|src1_sel|quad_perm|row_|wave_|bank_mask|row_mask|bound_ctrl|bitop3|byte_sel|fi:)")
SDST2 = re.compile(r"v_(add|sub|subrev|addc|subb|subbrev)_co_u32|v_mad_u64_u32|v_mad_i64_i32|v_div_scale_f(32|64)")
CARRY_IN = re.compile(r"v_(addc|subb|subbrev)_co_u32|v_cndmask_b32|v_div_fmas_f(32|64)")
SRC_BASE = [10, 14, 18]
DST_BASE = 30


def width(tok):
    m = re.fullmatch(r"[vs]\[(\d+):(\d+)\]", tok)
    if m:
        return int(m.group(2)) - int(m.group(1)) + 1
    if re.fullmatch(r"[vs]\d+|vcc_lo|vcc_hi|m0|exec_lo|exec_hi", tok):
        return 1
    if tok in ("vcc", "exec"):
        return 2
    return 0                                             # a constant / literal / special


def reg(base, w):
    return f"v{base}" if w == 1 else f"v[{base}:{base + w - 1}]"


def rewrite(text):
    """the instruction on the probe's registers, or (None, reason)"""
    t = text.strip()
    mn = t.split()[0]
    if SKIP.match(mn):
        return None, "left out by design"
    m = MOD.search(t)
    mods = t[m.start():] if m else ""
    body = t[:m.start()] if m else t
    ops = [o.strip() for o in body[len(mn):].split(",")] if body[len(mn):].strip() else []
    if not ops:
        return None, "no operands"
    out = []
    is_cmp = mn.startswith("v_cmp_")
    nsrc = 0
    seen = {}                                            # an operand register named twice stays ONE register (e.g. a horizontal add)
    for i, o in enumerate(ops):
        neg = o.startswith("-"); core = o[1:] if neg else o
        ab = core.startswith("|") and core.endswith("|"); core = core[1:-1] if ab else core
        w = width(core)
        if i == 0:                                       # destination
            if is_cmp:
                out.append("vcc"); continue
            if not core.startswith("v"):
                return None, f"destination {core}"
            if w > 4:
                return None, "wide destination"
            out.append(reg(DST_BASE, w)); continue
        if i == 1 and SDST2.match(mn) and (core.startswith("s") or core.startswith("vcc")):
            out.append("vcc"); continue                  # the carry / scale flag out
        if i == len(ops) - 1 and CARRY_IN.match(mn) and (core.startswith("s[") or core == "vcc") and not mn.startswith("v_div_fmas"):
            out.append("vcc"); continue                  # the carry / condition in
        if w == 0:
            r = core                                     # constant
        else:
            if core in seen:
                r = seen[core]
            else:
                if nsrc >= 3 or w > 4:
                    return None, "operand shape"
                r = seen[core] = reg(SRC_BASE[nsrc], w); nsrc += 1
        r = f"|{r}|" if ab else r
        out.append(("-" if neg else "") + r)
    new = f"{mn} {', '.join(out)}{mods}"
    if is_cmp:
        new += r"\n\ts_nop 1\n\tv_cndmask_b32_e64 v30, 0, 1, vcc"
    return new, None


def assembles(text):
    src = "\t" + text.replace(r"\n\t", "\n\t") + "\n"
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write(src)
    r = subprocess.run([str(LLVM / "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", f.name, "-o", "/dev/null"],
                       capture_output=True, text=True)
    Path(f.name).unlink()
    return r.returncode == 0, r.stderr.strip().split("\n")[0] if r.returncode else ""


def main():
    forms = {}
    keys = {}
    counts = {}
    left = {}
    with tempfile.TemporaryDirectory(prefix="forms_") as work:
        for p in guard.product_files():
            for co in guard.code_objects(p, work):
                dis = subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(co)], capture_output=True, text=True, check=True).stdout
                for line in dis.split("\n"):
                    t = line.split("//")[0].strip()
                    if not t.startswith("v_"):
                        continue
                    new, why = rewrite(t)
                    if new is None:
                        left[t.split()[0]] = why
                        continue
                    # one form per operand KINDS: literals and inline constants of one instruction are one datapath
                    mm = MOD.search(new)
                    operands, mods = (new[:mm.start()], new[mm.start():]) if mm else (new, "")
                    key = re.sub(r"(?<![\w\[:])-?(0x[0-9a-f]+|\d+(\.\d+)?)(?![\w\]:])", "K", operands) + mods
                    if key not in keys:
                        keys[key] = new
                    new = keys[key]
                    counts[new] = counts.get(new, 0) + 1
                    forms.setdefault(new, t)
    ok = []
    for new in sorted(forms, key=lambda k: -counts[k]):
        good, err = assembles(new)
        if good:
            ok.append(new)
        else:
            left[new] = "does not assemble on the probe's registers: " + err
    out = ROOT / "scripts" / "ubench" / "insn_probe_forms.inc"
    with open(out, "w") as f:
        f.write("// generated by scripts/gen_insn_probe_forms.py from the built product: every VALU instruction form it contains, on the\n"
                "// probe's registers (count in the library's code objects behind each).  X(index, text)\n#define PRODUCT_FORMS(X) \\\n")
        for i, t in enumerate(ok):
            f.write(f'    X({i}, "{t}") /* x{counts[t]} */ \\\n')
        f.write(f"\nconstexpr int N_PRODUCT_FORMS = {len(ok)};\n")
    print(f"{len(ok)} forms covering {sum(counts[t] for t in ok)} instructions -> {out}")
    for k, v in sorted(left.items()):
        print(f"  left out: {k}: {v}")


main()
