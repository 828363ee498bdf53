# rocprofv3 kernel-trace summary of the bench command itself.  prof_bench.sh [TAG=r03] [STEPS=40]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03}; STEPS=${2:-40}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python bench.py --steps $STEPS --warmup 8 --no-cpu-baseline --no-extras > gpurun_out/prof_bench.log 2>&1
find gpurun_out/prof_bench -name '*kernel_stats.csv' -exec cp {} gpurun_out/${TAG}_bench_n1_kernel_stats.csv \;
find gpurun_out/prof_bench -name '*kernel_trace.csv' -exec cp {} gpurun_out/ktrace.csv \;
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/ktrace.csv')))
print(len(rows), 'launches')
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
t0, t1 = iv[0][0], max(e for _, e in iv)
# busiest contiguous 40 % of the run = the timed loop: concurrency histogram there
evs = []
for s, e in iv:
    evs.append((s, 1)); evs.append((e, -1))
evs.sort()
W = 0.4 * (t1 - t0)
best = None
for k in range(0, 13):
    lo = t0 + k * 0.05 * (t1 - t0); hi = lo + W
    busy = sum(min(e, hi) - max(s, lo) for s, e in iv if e > lo and s < hi)
    if best is None or busy > best[0]: best = (busy, lo, hi)
_, lo, hi = best
conc = collections.Counter(); cur = 0; last = lo
for t, d in evs:
    if t > hi: break
    if t > lo: conc[cur] += t - max(last, lo)
    cur += d; last = max(t, lo)
tot = sum(conc.values())
print(f'densest 40 % of the trace ({(hi - lo) / 1e6:.0f} ms): kernels in flight (fraction of wall time)')
for k in sorted(conc): print(f'  {k}: {conc[k]/tot:.3f}')
print('sum kernel time / wall there =', best[0] / (hi - lo))
PY
rm -rf gpurun_out/prof_bench gpurun_out/ktrace.csv
grep -o '"value": [0-9.]*, "unit": "frames/s", "n_gpus"' gpurun_out/prof_bench.log | head -1
