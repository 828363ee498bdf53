# A/B of builds of libsslam_hip.so on ONE GPU box: ab_lib2.sh [pairs] "<flags A>" "<flags B>" ...
# Each variant is built in place (content-keyed objects), the batched LightGlue forward is profiled once
# with rocprofv3 --kernel-trace --stats and its per-kernel table printed; the default build is restored.
export SSLAM_EXPERIMENT_BUILD=1     # build.py refuses SSLAM_EXTRA_HIPCC_FLAGS without it
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B=$1; shift
i=0
for f in "$@"; do
  SSLAM_EXTRA_HIPCC_FLAGS="$f" python opencv-simpleslam_amd/build.py > /tmp/build_$i.log 2>&1 || { echo "build failed: $f"; tail -20 /tmp/build_$i.log; exit 1; }
  echo "=== variant $i: [$f]"
  bash scripts/prof_lg_batch.sh $B > /tmp/ab_$i.txt 2>&1
  head -8 /tmp/ab_$i.txt | cut -c1-150; tail -1 /tmp/ab_$i.txt
  tail -2 gpurun_out/prof_lgb.log
  cp gpurun_out/lgb_kernel_stats.csv gpurun_out/lgb_kernel_stats_v$i.csv
  i=$((i+1))
done
python opencv-simpleslam_amd/build.py > /dev/null 2>&1
