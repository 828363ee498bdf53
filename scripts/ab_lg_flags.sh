# A/B of LightGlue build flags on ONE box: ab_lg_flags.sh <kernel-name-pattern> "<flags A>" "<flags B>" ...
export SSLAM_EXPERIMENT_BUILD=1     # build.py refuses SSLAM_EXTRA_HIPCC_FLAGS without it
cd $GRAFT_REPO_ROOT
PAT=$1; shift
for fl in "$@"; do
  echo "=== flags: $fl"
  SSLAM_EXTRA_HIPCC_FLAGS="$fl" python opencv-simpleslam_amd/build.py > /tmp/ab_build.log 2>&1 || { tail -5 /tmp/ab_build.log; continue; }
  python scripts/time_lightglue_batch.py 2048 8 10 | tail -1
  bash scripts/prof_lg_batch.sh 8 | grep "$PAT\|LG total"
done
python opencv-simpleslam_amd/build.py > /dev/null 2>&1
