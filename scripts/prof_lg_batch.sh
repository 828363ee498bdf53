# per-kernel breakdown of the batched LightGlue forward: prof_lg_batch.sh [B=4]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B=${1:-4}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lgb -- python scripts/time_lightglue_batch.py 2048 $B 6 > gpurun_out/prof_lgb.log 2>&1
find gpurun_out/prof_lgb -name '*kernel_stats.csv' -exec cp {} gpurun_out/lgb_kernel_stats.csv \;
rm -rf gpurun_out/prof_lgb
python - $B <<'PY'
import csv, sys
B = int(sys.argv[1])
rows = [r for r in csv.DictReader(open('gpurun_out/lgb_kernel_stats.csv')) if 'lg_' in r['Name']]
calls = max(int(r['Calls']) for r in rows if 'lg_emit' in r['Name'])
tot = 0
for r in rows:
    per = int(r['TotalDurationNs']) / calls / 1e3 / B
    tot += per
    print(f"{r['Name'][22:95]:73s} x{int(r['Calls'])/calls:5.1f} {float(r['AverageNs'])/1e3:7.1f} us  per-pair {per:7.1f}")
print('LG total per pair (us)', tot)
PY
