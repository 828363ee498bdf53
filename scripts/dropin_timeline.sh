# GPU timeline of the drop-in `value` loop (one match per frame): where a frame's ~2.1 ms go - kernels back to back, or gaps?
#   dropin_timeline.sh [frames=16]      -> gpurun_out/dropin_timeline.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
N=${1:-16}
rm -rf gpurun_out/prof_dt
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_dt -- python scripts/dropin_bench.py $N > gpurun_out/prof_dt.log 2>&1
python - <<'PY' > gpurun_out/dropin_timeline.txt
import csv, glob
ev = []
for f in glob.glob('gpurun_out/prof_dt/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
for f in glob.glob('gpurun_out/prof_dt/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', r.get('Name', ''))))
ev.sort()
# frames of the planted `value` loop: an al_reset launch starts a frame's extraction; take frames whose chain holds exactly one
# lg_emit and one rs_tail (one match, one filter)
starts = [i for i, e in enumerate(ev) if 'al_reset_kernel' in e[2]]
frames = []
for a, b in zip(starts, starts[1:]):
    seg = ev[a:b]
    if sum('lg_emit' in e[2] for e in seg) == 1 and sum('rs_tail' in e[2] for e in seg) == 1 and sum('lg_prepare' in e[2] for e in seg) == 1:
        frames.append((a, b))
print(len(frames), 'single-match frames found')
import statistics
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n[:46]
rows = {}
tot_busy, tot_span, gaps_big = [], [], []
for a, b in frames[len(frames) // 3:]:
    seg = ev[a:b]
    # previous chain's end -> this frame's first event: the host turnaround; find the h2d copy before al_reset
    t0 = seg[0][0]
    end = max(e[1] for e in seg)
    nxt = ev[b][0]
    busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
    for s, e, n in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            if s - cur_e > 3000: gaps_big.append((s - cur_e, short(prev_n), short(n)))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        prev_n = n
    busy += cur_e - cur_s
    tot_busy.append(busy); tot_span.append(nxt - t0)
    rows.setdefault('frame period (al_reset to al_reset)', []).append(nxt - t0)
    rows.setdefault('GPU busy inside it (union of kernels + copies)', []).append(busy)
    rows.setdefault('last event end -> next al_reset (host turnaround incl. image upload)', []).append(nxt - end)
    k_al = [e for e in seg if e[2].startswith('void (anonymous namespace)::al_') or '::al_' in e[2] or 'al_' in e[2][:60] and 'lg_' not in e[2]]
    k_lg = [e for e in seg if 'lg_' in e[2]]
    k_rs = [e for e in seg if 'rs_' in e[2]]
    for nm, ks in (('ALIKED', k_al), ('LightGlue', k_lg), ('RANSAC', k_rs)):
        if ks:
            rows.setdefault(f'{nm}: first start -> last end', []).append(max(e[1] for e in ks) - min(e[0] for e in ks))
            rows.setdefault(f'{nm}: sum of kernel durations', []).append(sum(e[1] - e[0] for e in ks))
            rows.setdefault(f'{nm}: launches', []).append(len(ks) * 1000)
    if k_al and k_lg:
        rows.setdefault('ALIKED last end -> LightGlue first start', []).append(min(e[0] for e in k_lg) - max(e[1] for e in k_al))
    if k_lg and k_rs:
        rows.setdefault('LightGlue last end -> RANSAC first start', []).append(min(e[0] for e in k_rs) - max(e[1] for e in k_lg))
for k, v in rows.items():
    print(f'{k:75s} median {statistics.median(v) / 1e3:9.1f} us   (n = {len(v)})')
from collections import Counter
c = Counter((g[1], g[2]) for g in gaps_big)
print('gaps > 3 us inside a frame, by (kernel before, kernel after): count, median us')
for (p, n), cnt in c.most_common(25):
    vals = [g[0] for g in gaps_big if g[1] == p and g[2] == n]
    print(f'  {p:46s} -> {n:46s} x{cnt:3d}  {statistics.median(vals) / 1e3:7.1f}')
PY
rm -rf gpurun_out/prof_dt
cat gpurun_out/dropin_timeline.txt
