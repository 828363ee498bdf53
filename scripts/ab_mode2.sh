# A/B of linear-kernel modes of the batched forward on ONE box: ab_mode2.sh [pairs] mode mode ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B=$1; shift
for m in "$@"; do
  echo "=== SSLAM_BIG_GEMM=$m"
  SSLAM_BIG_GEMM=$m bash scripts/prof_lg_batch.sh $B > /tmp/ab_mode.txt 2>&1
  head -8 /tmp/ab_mode.txt | cut -c1-140; tail -1 /tmp/ab_mode.txt
  cp gpurun_out/lgb_kernel_stats.csv gpurun_out/lgb_kernel_stats_m$m.csv
  SSLAM_BIG_GEMM=$m python scripts/time_lightglue_batch.py 2048 $B 10 | tail -1
done
