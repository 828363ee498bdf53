# frames/s of the bench pipeline for two linear-kernel modes, interleaved on ONE box
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for m in "$@"; do
    SSLAM_BIG_GEMM=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('mode=$m', d['value'], 'fps; structured', d['structured_input']['value'], 'lg batch ms', d['roofline']['lightglue_batch_ms_isolated'], flush=True)"
  done
done
