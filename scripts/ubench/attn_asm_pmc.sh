# SQ counters + effective clock of the attention kernels in the stand-alone bench (hs, asm, asm MFMA-only)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc scripts/ubench/attn_bench.hip -o /tmp/attn_b 2>/dev/null
mk() { ATTN_ASM_ABL=$1 python3 opencv-simpleslam_amd/csrc/gen_lg_attention_asm.py > /tmp/$2.s; /opt/rocm/lib/llvm/bin/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/$2.s -o /tmp/$2.o; /opt/rocm/lib/llvm/bin/ld.lld -shared /tmp/$2.o -o /tmp/$2.hsaco; }
mk "" full; mk novalu,nods,nodma mfma; mk nomfma rest; mk novalu noval
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
run() { # tag, ATTN_PP, hsaco
  rm -rf /tmp/pmc_$1
  ATTN_PP=$2 ATTN_HSACO=$3 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$1 -- /tmp/attn_b 2048 8 1 3 > /tmp/pmc_$1.log 2>&1
  python3 - $1 <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
cc = glob.glob(f"/tmp/pmc_{tag}/**/*counter_collection.csv", recursive=True)
kt = glob.glob(f"/tmp/pmc_{tag}/**/*kernel_trace.csv", recursive=True)
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(cc[0])):
    if "attention" not in r["Kernel_Name"]: continue
    a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0])) if "attention" in r["Kernel_Name"]]
dur.sort(); d = dur[len(dur) // 2] / 1e3
v = {k: a[0] / a[1] for k, a in acc.items()}
wc = v["SQ_WAVE_CYCLES"]
print(f"[{tag}] {d:.1f} us | clock {v['GRBM_GUI_ACTIVE'] / 8 / d / 1e3:.2f} GHz | wave-cycles(quad) {wc:.3g}: active {v['SQ_ACTIVE_INST_ANY'] / wc:.2f} (valu {v['SQ_ACTIVE_INST_VALU'] / wc:.2f}) issue-stall {v['SQ_WAIT_INST_ANY'] / wc:.2f} wait {v['SQ_WAIT_ANY'] / wc:.2f} | VALU insts {v['SQ_INSTS_VALU']:.3g} | MFMA busy {v['SQ_VALU_MFMA_BUSY_CYCLES']:.3g} = {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 1024):.2f} of SIMD-cycles | SQ busy {v['SQ_BUSY_CYCLES']:.3g}")
PY
}
run hs 3 /tmp/full.hsaco
run asm 4 /tmp/full.hsaco
run asm_mfma 4 /tmp/mfma.hsaco
run asm_novalu 4 /tmp/noval.hsaco
run asm_rest 4 /tmp/rest.hsaco
