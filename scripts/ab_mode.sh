# A/B of the linear-kernel modes of the batched forward on ONE box: ab_mode.sh "<mode A>" "<mode B>" [pairs=8]
# (SSLAM_BIG_GEMM: 0 = ring kernels, 1 = batched form)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B=${3:-8}
for m in "$1" "$2" "$1" "$2"; do
  echo "=== SSLAM_BIG_GEMM=$m"
  SSLAM_BIG_GEMM=$m bash scripts/prof_lg_batch.sh $B > /tmp/ab_mode.txt 2>&1
  head -7 /tmp/ab_mode.txt | cut -c1-125; tail -1 /tmp/ab_mode.txt
  SSLAM_BIG_GEMM=$m python scripts/time_lightglue_batch.py 2048 $B 10 | tail -1
done
