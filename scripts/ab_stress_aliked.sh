# which build switch makes ALIKED non-deterministic under concurrency: ab_stress_aliked.sh <repeats> "<flags A>" "<flags B>" ...
export SSLAM_EXPERIMENT_BUILD=1     # build.py refuses SSLAM_EXTRA_HIPCC_FLAGS without it
cd $GRAFT_REPO_ROOT
R=$1; shift
for fl in "$@"; do
  echo "=== flags: $fl"
  SSLAM_EXTRA_HIPCC_FLAGS="$fl" python opencv-simpleslam_amd/build.py > /tmp/ab_build.log 2>&1 || { tail -5 /tmp/ab_build.log; continue; }
  timeout -k 10 400 python scripts/stress_aliked_repeat.py $R 3 2 2>&1 | tail -4
done
python opencv-simpleslam_amd/build.py > /dev/null 2>&1
