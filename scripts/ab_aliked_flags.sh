# A/B of ALIKED build flags on ONE box: ab_aliked_flags.sh <kernel-name-pattern> "<flags A>" "<flags B>" ...   (each a SSLAM_EXTRA_HIPCC_FLAGS value)
export SSLAM_EXPERIMENT_BUILD=1     # build.py refuses SSLAM_EXTRA_HIPCC_FLAGS without it
cd $GRAFT_REPO_ROOT
PAT=$1; shift
for fl in "$@"; do
  echo "=== flags: $fl"
  SSLAM_EXTRA_HIPCC_FLAGS="$fl" python opencv-simpleslam_amd/build.py > /tmp/ab_build.log 2>&1 || { tail -5 /tmp/ab_build.log; continue; }
  python scripts/time_aliked.py 20 8 0
  python scripts/time_aliked.py 20 1 0
  bash scripts/prof_aliked.sh 8 | grep "$PAT\|kernel time per frame"
done
python opencv-simpleslam_amd/build.py > /dev/null 2>&1
