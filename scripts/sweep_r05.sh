# r05: the pipeline's stream / batch configuration again after the precision default and the assignment changes
#   (SSLAM_BENCH_NM matcher streams, SSLAM_BENCH_EF frames per extractor call, SSLAM_BENCH_PAIRS pairs per launch, SSLAM_BENCH_FRAMES per round)
cd $GRAFT_REPO_ROOT
for cfg in "3 8 8 24" "2 8 8 24" "4 8 8 24" "3 4 8 24" "3 8 12 24" "2 8 12 24" "3 8 8 32" "3 8 16 32" "2 8 16 32"; do
  set -- $cfg
  SSLAM_BENCH_NM=$1 SSLAM_BENCH_EF=$2 SSLAM_BENCH_PAIRS=$3 SSLAM_BENCH_FRAMES=$4 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('NM=$1 EF=$2 PAIRS=$3 FRAMES=$4:', d['value'], 'frames/s, ms_per_step', d['ms_per_step'])"
done
