# stream / batch configuration sweep of the bench pipeline: "FRAMES NE NM PAIRS" per entry
cd $GRAFT_REPO_ROOT
for cfg in ${CFGS:-"24 2 2 8" "32 2 2 8" "16 2 2 8" "48 2 2 8" "32 2 2 16" "40 2 2 10" "24 2 2 12" "32 3 2 8"}; do
  set -- $cfg
  SSLAM_BENCH_FRAMES=$1 SSLAM_BENCH_NE=$2 SSLAM_BENCH_NM=$3 SSLAM_BENCH_PAIRS=$4 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('B=$1 NE=$2 NM=$3 P=$4', d['value'], 'fps; structured', d['structured_input']['value'], 'lg batch ms', d['roofline']['lightglue_batch_ms_isolated'], 'attn us', d['roofline']['avg_launch_us'], flush=True)"
done
