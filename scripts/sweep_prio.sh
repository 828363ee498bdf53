for cfg in "0 0" "-1 0" "0 -1" "0 0"; do
  set -- $cfg
  SSLAM_BENCH_PRIO_E=$1 SSLAM_BENCH_PRIO_M=$2 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('prio_e=$1 prio_m=$2', d['value'], 'fps; structured', d['structured_input']['value'])"
done
