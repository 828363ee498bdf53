# SQ counters per launch of the ALIKED kernels at F frames per launch sequence.  pmc_aliked.sh [F=8]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
F=${1:-8}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_al -- python scripts/time_aliked.py 4 $F > gpurun_out/pmc_al.log 2>&1
find gpurun_out/pmc_al -name '*counter_collection.csv' -exec cp {} gpurun_out/pmc_al.csv \;
find gpurun_out/pmc_al -name '*kernel_trace.csv' -exec cp {} gpurun_out/pmc_al_trace.csv \;
rm -rf gpurun_out/pmc_al
python - <<'PY'
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open("gpurun_out/pmc_al.csv")):
    k = r["Kernel_Name"]
    if "al_" not in k: continue
    a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
dur = collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/pmc_al_trace.csv")):
    dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for k, d in acc.items():
    v = {c: a[0] / a[1] for c, a in d.items()}
    wc = v.get("SQ_WAVE_CYCLES", 1)
    us = sorted(dur[k])[len(dur[k]) // 2] / 1e3
    rows.append((us, k, v, wc))
for us, k, v, wc in sorted(rows, reverse=True)[:16]:
    print(f"{k[22:80]:58s} {us:7.1f} us  active {v['SQ_ACTIVE_INST_ANY']/wc:.2f} (valu {v['SQ_ACTIVE_INST_VALU']/wc:.2f}) stall {v['SQ_WAIT_INST_ANY']/wc:.2f} wait {v['SQ_WAIT_ANY']/wc:.2f} | VALU {v['SQ_INSTS_VALU']:.3g} VMEM_RD {v['SQ_INSTS_VMEM_RD']:.3g} LDS {v['SQ_INSTS_LDS']:.3g}")
PY
