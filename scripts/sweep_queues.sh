# GPU_MAX_HW_QUEUES: how many hardware queues the HIP runtime multiplexes this process's streams onto
for cfg in "2 1 7" "3 1 7" "4 1 7" "5 1 7" "6 1 7" "4 1 11" "4 2 10" "3 1 8" "4 1 7"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 SSLAM_BENCH_NE=$2 SSLAM_BENCH_NM=$3 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('GPU_MAX_HW_QUEUES=$1 NE=$2 NM=$3', d['value'], 'fps; structured', d['structured_input']['value'])"
done
