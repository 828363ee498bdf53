# stream / batch configuration sweep of the bench pipeline: "FRAMES NE NM PAIRS" per entry
cd $GRAFT_REPO_ROOT
for cfg in "24 2 2 8" "24 1 2 8" "24 2 1 8" "24 2 3 8" "24 2 2 12" "32 2 2 16" "32 2 1 16" "24 2 2 6" "24 2 2 4" "24 3 2 8"; do
  set -- $cfg
  SSLAM_BENCH_FRAMES=$1 SSLAM_BENCH_NE=$2 SSLAM_BENCH_NM=$3 SSLAM_BENCH_PAIRS=$4 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('B=$1 NE=$2 NM=$3 P=$4', d['value'], 'fps; structured', d['structured_input']['value'], 'lg batch ms', d['roofline']['lightglue_batch_ms_isolated'], 'attn us', d['roofline']['avg_launch_us'], flush=True)"
done
