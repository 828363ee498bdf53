# per-kernel profile of ALIKED extraction: prof_aliked.sh [F=1]  (F frames per launch sequence)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
F=${1:-1}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_al -- python scripts/time_aliked.py 20 $F > gpurun_out/prof_al.log 2>&1
find gpurun_out/prof_al -name '*kernel_stats.csv' -exec cp {} gpurun_out/aliked_kernel_stats_F$F.csv \;
rm -rf gpurun_out/prof_al
python - $F <<'PY'
import csv, sys
F = int(sys.argv[1])
rows = list(csv.DictReader(open(f'gpurun_out/aliked_kernel_stats_F{F}.csv')))
rows = [r for r in rows if 'al_' in r['Name']]
calls = max(int(r['Calls']) for r in rows if 'al_aggregate' in r['Name'])
tot = 0; launches = 0
for r in rows:
    per = int(r['TotalDurationNs']) / calls / 1e3 / F
    tot += per; launches += int(r['Calls']) / calls
    print(f"{r['Name'][22:100]:78s} x{int(r['Calls'])/calls:4.1f} {float(r['AverageNs'])/1e3:8.1f} us  per-frame {per:7.1f}")
print(f'F = {F}: kernel time per frame {tot:.1f} us, {launches:.0f} launches per call = {launches / F:.1f} per frame')
PY
tail -2 gpurun_out/prof_al.log
