# HBM-side traffic per launch of the LightGlue kernels (two separate PMC passes, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python scripts/time_lightglue.py 2048 2 > gpurun_out/pmc_$c.log 2>&1
  find gpurun_out/pmc_$c -name '*counter_collection.csv' -exec cp {} gpurun_out/pmc_$c.csv \;
  rm -rf gpurun_out/pmc_$c
done
python - <<'PY'
import csv, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f"gpurun_out/pmc_{c}.csv")):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"]
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for k, (v, n) in acc.items():
        out.setdefault(k, {})[c] = (v / n, n)
rows = []
for k, d in out.items():
    if "lg_" not in k:
        continue
    f, nf = d.get("FETCH_SIZE", (0, 0)); w, nw = d.get("WRITE_SIZE", (0, 0))
    # counters are in KiB; gfx950 correction: FETCH_SIZE tallies 128-B requests at 64 B -> x2
    rows.append((k[:70], nf, f * 1024 * 2 / 1e6, w * 1024 / 1e6))
rows.sort(key=lambda r: -(r[2] + r[3]))
with open("gpurun_out/r01_pmc_traffic_v6.csv", "w") as fh:
    fh.write("kernel,launches,fetch_MB_per_launch_corrected_x2,write_MB_per_launch\n")
    for r in rows:
        fh.write(f"\"{r[0]}\",{r[1]},{r[2]:.3f},{r[3]:.3f}\n")
        print(f"{r[0]:70s} n={r[1]:4d} fetch {r[2]:8.2f} MB  write {r[3]:8.2f} MB per launch")
PY
