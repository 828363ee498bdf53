# Gaps between consecutive kernels of the single-frame ALIKED sequence, plain launches and as a cached hipGraph.   gaps_aliked.sh [F=1]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
F=${1:-1}
for G in 0 1; do
  rm -rf gpurun_out/prof_gap
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gap -- python scripts/time_aliked.py 30 $F $G > gpurun_out/prof_gap.log 2>&1
  tail -1 gpurun_out/prof_gap.log
  python - $G <<'PY'
import csv, glob, statistics, sys, collections
ev = []
for f in glob.glob('gpurun_out/prof_gap/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')))
ev.sort()
starts = [i for i, e in enumerate(ev) if 'al_reset_kernel' in e[2]]
gaps = collections.defaultdict(list); spans = []; sums = []
for a, b in list(zip(starts, starts[1:]))[10:]:
    seg = ev[a:b]
    spans.append((seg[-1][1] - seg[0][0]) / 1e3); sums.append(sum(e[1] - e[0] for e in seg) / 1e3)
    for k, (x, y) in enumerate(zip(seg, seg[1:])):
        gaps[(k, x[2][:40], y[2][:40])].append((y[0] - x[1]) / 1e3)
print(f"graphs={sys.argv[1]}: calls {len(spans)}, first start -> last end median {statistics.median(spans):.1f} us, sum of kernel durations {statistics.median(sums):.1f} us, launches {len(ev[starts[10]:starts[11]])}")
for (k, a, b), v in sorted(gaps.items()):
    m = statistics.median(v)
    if m > 1.5:
        print(f"   after launch {k:2d}: {a:40s} -> {b:40s} gap median {m:6.1f} us (n {len(v)})")
PY
done
rm -rf gpurun_out/prof_gap
