# HBM-side traffic per launch of the ALIKED kernels at F frames per launch sequence (two separate PMC passes, as
# MI355X_MICROARCH.md prescribes).   pmc_traffic_aliked.sh [TAG=r04] [F=8]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r04}; F=${2:-8}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python scripts/time_aliked.py 4 $F > gpurun_out/pmc_$c.log 2>&1
  find gpurun_out/pmc_$c -name '*counter_collection.csv' -exec cp {} gpurun_out/pmc_$c.csv \;
  rm -rf gpurun_out/pmc_$c
done
python - $TAG $F <<'PY'
import csv, collections, sys
tag, F = sys.argv[1], int(sys.argv[2])
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f"gpurun_out/pmc_{c}.csv")):
        if r["Counter_Name"] != c:
            continue
        acc[r["Kernel_Name"]][0] += float(r["Counter_Value"]); acc[r["Kernel_Name"]][1] += 1
    for k, (v, n) in acc.items():
        out.setdefault(k, {})[c] = (v / n, n)
rows = []
for k, d in out.items():
    if "al_" not in k:
        continue
    f, nf = d.get("FETCH_SIZE", (0, 0)); w, nw = d.get("WRITE_SIZE", (0, 0))
    # counters are in KiB; gfx950 correction: FETCH_SIZE tallies 128-B requests at 64 B -> x2
    rows.append((k[:100], nf, f * 1024 * 2 / 1e6 / F, w * 1024 / 1e6 / F))
rows.sort(key=lambda r: -(r[2] + r[3]))
with open(f"gpurun_out/{tag}_pmc_traffic_aliked.csv", "w") as fh:
    fh.write(f"# ALIKED 1241 x 376 -> 2048 keypoints, {F} frames per launch; MB per FRAME\n")
    fh.write("kernel,launches,fetch_MB_per_frame_corrected_x2,write_MB_per_frame\n")
    for r in rows:
        fh.write(f"\"{r[0]}\",{r[1]},{r[2]:.2f},{r[3]:.2f}\n")
        print(f"{r[0][22:100]:78s} n={r[1]:3d} fetch {r[2]:7.2f} MB  write {r[3]:7.2f} MB per frame")
print("sum per frame: fetch %.1f MB, write %.1f MB" % (sum(r[2] for r in rows), sum(r[3] for r in rows)))
PY
