export SSLAM_EXPERIMENT_BUILD=1     # build.py refuses SSLAM_EXTRA_HIPCC_FLAGS without it
cd $GRAFT_REPO_ROOT
for fl in "-DAL_DCN4_CS=2" "-DAL_DCN4_CS=4"; do
  echo "=== $fl"
  SSLAM_EXTRA_HIPCC_FLAGS="$fl" python opencv-simpleslam_amd/build.py > /tmp/ab_build.log 2>&1 || { tail -5 /tmp/ab_build.log; continue; }
  python scripts/time_aliked.py 30 8 0; python scripts/time_aliked.py 30 1 0
  bash scripts/prof_aliked.sh 8 | grep "dcn_h\|per frame" | cut -c1-120
done
python opencv-simpleslam_amd/build.py > /dev/null 2>&1
