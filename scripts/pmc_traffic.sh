# HBM-side traffic per launch of the LightGlue kernels (two separate PMC passes, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
#   pmc_traffic.sh [TAG=r02] [B=8]     batched forward of B pairs at 2048 x 2048
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r02}; B=${2:-8}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python scripts/time_lightglue_batch.py 2048 $B 2 > gpurun_out/pmc_$c.log 2>&1
  find gpurun_out/pmc_$c -name '*counter_collection.csv' -exec cp {} gpurun_out/pmc_$c.csv \;
  rm -rf gpurun_out/pmc_$c
done
python - $TAG $B <<'PY'
import csv, collections, json, sys
tag, B = sys.argv[1], int(sys.argv[2])
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
    rows.append((k[:90], nf, f * 1024 * 2 / 1e6, w * 1024 / 1e6))
rows.sort(key=lambda r: -(r[2] + r[3]))
with open(f"gpurun_out/{tag}_pmc_traffic.csv", "w") as fh:
    fh.write(f"# batched LightGlue forward, {B} pairs of 2048 x 2048 per launch\n")
    fh.write("kernel,launches,fetch_MB_per_launch_corrected_x2,write_MB_per_launch\n")
    for r in rows:
        fh.write(f"\"{r[0]}\",{r[1]},{r[2]:.3f},{r[3]:.3f}\n")
        print(f"{r[0]:90s} n={r[1]:4d} fetch {r[2]:8.2f} MB  write {r[3]:8.2f} MB per launch")
for r in rows:
    if "lg_attention_asm" in r[0]:                       # (lg_attention_asm_kernel = f16x3, lg_attention_asm_p1_kernel = f16x3p1, the default)
        K, NI = 2048, 2 * B
        json.dump({"kernel": r[0].strip('"'), "workload": f"{B} pairs of 2048 x 2048 per launch, 4 heads x 64, no key split",
                   "fetch_bytes_per_launch": int(r[2] * 1e6), "write_bytes_per_launch": int(r[3] * 1e6),
                   "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace "
                             "(scripts/pmc_traffic.sh); FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md; counters in KiB",
                   "algorithmic_bytes_per_launch": {"read_q_k_vT_planes": NI * K * 256 * 2 * 2 * 3,
                                                    "write_context_planes": NI * K * 256 * 2 * 2},
                   "source": f"profiles/{tag}_pmc_traffic.csv"}, open(f"gpurun_out/{tag}_attention_traffic.json", "w"), indent=1)
PY
