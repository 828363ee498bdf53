# SQ counters per launch for the LightGlue kernels (one PMC pass, 8 SQ slots).  pmc_sq.sh [TAG=r02] [B=8]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r02}; B=${2:-8}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_sq -- python scripts/time_lightglue_batch.py 2048 $B 2 > gpurun_out/pmc_sq.log 2>&1
find gpurun_out/pmc_sq -name '*counter_collection.csv' -exec cp {} gpurun_out/pmc_sq.csv \;
rm -rf gpurun_out/pmc_sq
python - $TAG $B <<'PY'
import csv, collections, sys
tag, B = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open("gpurun_out/pmc_sq.csv")):
    k = r["Kernel_Name"]
    if "lg_" not in k: continue
    a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
names = ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_VALU_MFMA_BUSY_CYCLES"]
with open(f"gpurun_out/{tag}_pmc_sq.csv", "w") as fh:
    fh.write(f"# batched LightGlue forward, {B} pairs of 2048 x 2048 per launch\n")
    fh.write("kernel,launches," + ",".join(n + "_per_launch" for n in names) + "\n")
    for k, d in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"][0]):
        n = d["SQ_WAVE_CYCLES"][1]
        vals = [d[c][0] / max(d[c][1], 1) for c in names]
        fh.write('"' + k[:90] + '",' + str(n) + "," + ",".join(f"{v:.0f}" for v in vals) + "\n")
        if "attention" in k or "linear" in k:
            wc = vals[0]
            print(f"{k[22:90]:68s} active {vals[1]/wc:.2f} issue-stall {vals[2]/wc:.2f} wait {vals[3]/wc:.2f} | VALU {vals[4]:.0f} MFMA {vals[5]:.0f} LDS {vals[6]:.0f} | mfma busy cycles {vals[7]:.0f}")
PY
