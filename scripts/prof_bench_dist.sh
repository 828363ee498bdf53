# kernel-trace summary of bench.py in the forced-distributed mode (RCCL, one rank): what the collation path adds
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
SSLAM_BENCH_FORCE_DIST=1 SSLAM_DIST_BACKEND=nccl rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dist -- python bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --no-extras > gpurun_out/prof_dist.log 2>&1
f=$(find gpurun_out/prof_dist -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/bench_dist_kernel_stats.csv
rm -rf gpurun_out/prof_dist
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/bench_dist_kernel_stats.csv")))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"]))[:14]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "%9.1f us" % (float(r["AverageNs"]) / 1e3), "%5.1f %%" % (100 * int(r["TotalDurationNs"]) / tot))
PY
grep -o '"value": [0-9.]*' gpurun_out/prof_dist.log | head -1
