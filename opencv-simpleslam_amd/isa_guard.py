"""Static scan of gfx950 code for the instruction form that profiles/r06_aggregate_rnorm_diagnosis.md (section 5) found to compute
its LOW half with src1's high half read as 0.0 in lanes 48..63, while another wave on the SIMD executes an MFMA with 128-bit or
wider A / B operands (v_mfma_*_16x16x32_f16 / bf16, 32x32x16_f16 / bf16, 32x32x32_i8, f8f6f4):

    v_pk_{mul,add,fma}_f32  D, S0, S1 [, S2]  op_sel:[0,1(,x)]      with S0 != S1

(whatever op_sel_hi and op_sel[2] are; the same instruction with S0 == S1 - a horizontal add - never failed in 5e9 executions, nor
did any other op_sel).  The compiler's SLP vectoriser forms it freely; the product must not contain it.

usage: python opencv-simpleslam_amd/isa_guard.py [FILE ...]     FILE: a shared library / object with HIP fat binaries, a code object (.hsaco / .co), or
                                        assembly text (.s); default: the built product library and the code objects beside it
Per kernel: packed fp32 instructions, those of the failing form, same-source ones, wide-operand MFMAs.  Exit code 1 if the
failing form occurs.  build.py runs check() on every product library it links and refuses the build; tests/test_isa_guard.py
checks the scanner and the built product.  (scripts/scan_pk_opsel.py is the same command line.)"""
import collections, re, shutil, subprocess, sys, tempfile
from pathlib import Path

PKG = Path(__file__).resolve().parent
LLVM = Path("/opt/rocm/lib/llvm/bin")
PK = re.compile(r"v_pk_(mul|add|fma)_f32\s+(.*)")
SEL = re.compile(r"op_sel:\[(\d),(\d)")
# MFMAs whose A / B operands are four or more VGPRs
WIDE = re.compile(r"v_mfma_\w+?_(32x32x16|16x16x32)_(f16|bf16)\b|v_mfma_\w*f8f6f4|v_mfma_i32_(32x32x32|16x16x64)_i8|v_smfmac_")
LABEL = re.compile(r"^(?:[0-9a-f]+ )?<?([A-Za-z_$][\w$.]*)>?:\s*(?:;.*)?$")


def classify(text):
    """'' (not packed fp32) | 'pk' | 'same' (op_sel:[0,1], S0 == S1) | 'bad' (op_sel:[0,1], S0 != S1)"""
    t = text.split("//")[0].split(";")[0].strip()
    m = PK.match(t)
    if not m:
        return ""
    s = SEL.search(t)
    if not (s and s.group(1) == "0" and s.group(2) == "1"):
        return "pk"
    ops = [o.strip() for o in re.sub(r"\s+(op_sel|op_sel_hi|neg_lo|neg_hi|clamp)\b.*", "", m.group(2)).split(",")]
    return "same" if len(ops) >= 3 and ops[1] == ops[2] else "bad"


def scan_text(lines):
    """{kernel: dict(pk=, bad=, same=, wide=, bad_text=[...])} of an assembly listing or a disassembly"""
    res = collections.OrderedDict()
    cur = "?"
    for line in lines:
        lab = LABEL.match(line.strip()) if not line.startswith(("\t", " ")) or line.strip().endswith(">:") else None
        if lab and not lab.group(1).startswith((".L", "BB")):
            cur = lab.group(1)
            continue
        t = line.strip()
        c = classify(t)
        if c or WIDE.match(t):
            r = res.setdefault(cur, dict(pk=0, bad=0, same=0, wide=0, bad_text=[]))
            if c:
                r["pk"] += 1
                if c in ("bad", "same"):
                    r[c] += 1
                if c == "bad":
                    r["bad_text"].append(t.split("//")[0].strip())
            else:
                r["wide"] += 1
    return res


def code_objects(path, work):
    """the gfx950 code objects inside `path` (a HIP shared library or object: its fat binaries; a code object: itself)"""
    path = Path(path)
    head = path.read_bytes()[:20]
    is_elf = head[:4] == b"\x7fELF"
    if is_elf and head[18:20] == (224).to_bytes(2, "little"):          # e_machine EM_AMDGPU
        return [path]
    tmp = Path(work) / path.name
    shutil.copy(path, tmp)
    subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", tmp.name], cwd=work, capture_output=True, check=True)
    return sorted(Path(work).glob(tmp.name + ".*gfx950*"))


def scan(paths):
    """{file: {kernel: counts}} over libraries / code objects / assembly files"""
    out = collections.OrderedDict()
    with tempfile.TemporaryDirectory(prefix="pkscan_") as work:
        for p in paths:
            p = Path(p)
            if p.suffix == ".s":
                out[p.name] = scan_text(open(p))
                continue
            merged = collections.OrderedDict()
            for co in code_objects(p, work):
                dis = subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(co)], capture_output=True, text=True, check=True).stdout
                for k, v in scan_text(dis.split("\n")).items():
                    merged[k] = v
            out[p.name] = merged
    return out


def failing(result):
    """[(file, kernel, instruction)] of the failing form"""
    return [(f, k, t) for f, ks in result.items() for k, v in ks.items() for t in v["bad_text"]]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def check(paths):
    """raises RuntimeError naming every instruction of the failing form in `paths`"""
    bad = failing(scan(paths))
    if bad:
        names = demangle(sorted({k for _, k, _ in bad}))
        lines = [f"  {f}: {names[k][:120]}: {t}" for f, k, t in bad[:20]]
        raise RuntimeError(
            "packed fp32 instructions of the form `op_sel:[0,1]` with two different sources - their low half is computed with src1's high "
            "half read as 0.0 in lanes 48..63 while another wave executes a wide-operand MFMA (profiles/r06_aggregate_rnorm_diagnosis.md, "
            f"section 5) - {len(bad)} in this build:\n" + "\n".join(lines) + "\nGive the kernel "
            '__attribute__((target("no-packed-fp32-ops"))) (as al_aggregate_kernel has) or change the expression the vectoriser pairs.')


def product_files():
    lib = PKG / "lib"
    return [lib / "libsslam_hip.so"] + sorted((lib / "obj").glob("*.hsaco"))


def main():
    paths = sys.argv[1:] or product_files()
    res = scan(paths)
    for f, ks in res.items():
        names = demangle(list(ks))
        print(f"{f}:")
        for k, v in ks.items():
            flag = "   <-- FAILING FORM" if v["bad"] else ""
            print(f"  packed fp32 {v['pk']:4d}  op_sel:[0,1] S0!=S1 {v['bad']:3d}  S0==S1 {v['same']:2d}  wide-operand MFMA {v['wide']:4d}  {names[k][:100]}{flag}")
            for t in v["bad_text"][:4]:
                print(f"        {t}")
    bad = failing(res)
    print(f"instructions of the failing form: {len(bad)}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
