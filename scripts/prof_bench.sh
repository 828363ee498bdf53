# rocprofv3 kernel-trace summary of the bench command itself.  prof_bench.sh [TAG=r02]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r02}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_bench.log 2>&1
find gpurun_out/prof_bench -name '*kernel_stats.csv' -exec cp {} gpurun_out/${TAG}_bench_n1_kernel_stats.csv \;
find gpurun_out/prof_bench -name '*kernel_trace.csv' -exec cp {} gpurun_out/ktrace.csv \;
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/ktrace.csv')))
print(len(rows), 'launches')
t0 = min(int(r['Start_Timestamp']) for r in rows); t1 = max(int(r['End_Timestamp']) for r in rows)
evs = []
for r in rows:
    evs.append((int(r['Start_Timestamp']), 1)); evs.append((int(r['End_Timestamp']), -1))
evs.sort()
conc = collections.Counter(); cur = 0; last = evs[0][0]
lo = t0 + 0.5 * (t1 - t0)
for t, d in evs:
    if t > lo: conc[cur] += t - max(last, lo)
    cur += d; last = t
tot = sum(conc.values())
print('concurrency histogram (fraction of wall time, second half of the run):')
for k in sorted(conc): print(f'  {k} kernels in flight: {conc[k]/tot:.3f}')
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows if int(r['Start_Timestamp']) > lo)
print('sum kernel time / wall =', busy / (t1 - lo))
PY
rm -rf gpurun_out/prof_bench gpurun_out/ktrace.csv
tail -2 gpurun_out/prof_bench.log | cut -c1-400
