cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_al -- python scripts/time_aliked.py > gpurun_out/prof_al.log 2>&1
find gpurun_out/prof_al -name '*kernel_stats.csv' -exec cp {} gpurun_out/aliked_kernel_stats.csv \;
rm -rf gpurun_out/prof_al
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/aliked_kernel_stats.csv')))
rows = [r for r in rows if 'al_' in r['Name']]
calls = max(int(r['Calls']) for r in rows if 'al_aggregate' in r['Name'])
tot = 0
for r in rows:
    per = int(r['TotalDurationNs']) / calls / 1e3
    tot += per
    print(f"{r['Name'][22:100]:78s} x{int(r['Calls'])/calls:4.1f} {float(r['AverageNs'])/1e3:8.1f} us  per-frame {per:7.1f}")
print('total per frame', tot)
PY
tail -2 gpurun_out/prof_al.log
