cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lg -- python scripts/time_lightglue.py 2048 10 > gpurun_out/prof_lg.log 2>&1
find gpurun_out/prof_lg -name '*kernel_stats.csv' -exec cp {} gpurun_out/lg_kernel_stats.csv \;
rm -rf gpurun_out/prof_lg
python - <<'PY'
import csv
rows = [r for r in csv.DictReader(open('gpurun_out/lg_kernel_stats.csv')) if 'lg_' in r['Name']]
calls = max(int(r['Calls']) for r in rows if 'lg_emit' in r['Name'])
tot = 0
for r in rows:
    per = int(r['TotalDurationNs']) / calls / 1e3
    tot += per
    print(f"{r['Name'][22:95]:73s} x{int(r['Calls'])/calls:5.1f} {float(r['AverageNs'])/1e3:7.1f} us  per-pair {per:7.1f}")
print('LG total per pair', tot)
PY
