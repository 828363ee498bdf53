# Gaps between consecutive kernels of a launch sequence under rocprofv3 --kernel-trace.
#   gaps_generic.sh <first-kernel-substring> <python script> [args...]      (env passes through, e.g. SSLAM_GRAPHS=1)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
FIRST=$1; shift
rm -rf gpurun_out/prof_gap
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gap -- python "$@" > gpurun_out/prof_gap.log 2>&1
tail -1 gpurun_out/prof_gap.log
python - "$FIRST" <<'PY'
import csv, glob, statistics, sys, collections
ev = []
for f in glob.glob('gpurun_out/prof_gap/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')))
ev.sort()
starts = [i for i, e in enumerate(ev) if sys.argv[1] in e[2]]
gaps = collections.defaultdict(list); spans = []; sums = []; n = []
for a, b in list(zip(starts, starts[1:]))[len(starts) // 2:]:
    seg = ev[a:b]
    spans.append((seg[-1][1] - seg[0][0]) / 1e3); sums.append(sum(e[1] - e[0] for e in seg) / 1e3); n.append(len(seg))
    for k, (x, y) in enumerate(zip(seg, seg[1:])):
        gaps[(k, x[2][:40], y[2][:40])].append((y[0] - x[1]) / 1e3)
print(f"sequences {len(spans)}, launches {statistics.median(n)}, first start -> last end median {statistics.median(spans):.1f} us, sum of kernel durations {statistics.median(sums):.1f} us")
for (k, a, b), v in sorted(gaps.items()):
    m = statistics.median(v)
    if m > 1.5 and len(v) > len(spans) // 2:
        print(f"   after launch {k:2d}: {a:40s} -> {b:40s} gap median {m:6.1f} us (n {len(v)})")
PY
rm -rf gpurun_out/prof_gap
