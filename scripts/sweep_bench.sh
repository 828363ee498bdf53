for cfg in "24 1 7" "28 1 7" "56 1 7" "24 2 6" "28 1 7" "42 1 7"; do
  set -- $cfg
  SSLAM_BENCH_FRAMES=$1 SSLAM_BENCH_NE=$2 SSLAM_BENCH_NM=$3 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('B=$1 NE=$2 NM=$3', d['value'], 'fps; structured', d['structured_input']['value'])"
done
