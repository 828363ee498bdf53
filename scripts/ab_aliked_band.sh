# A/B of the XCD-aware tile placement of ALIKED's halo kernels: per-kernel time and HBM-side traffic, placement on / off.
export SSLAM_EXPERIMENT_BUILD=1
cd $GRAFT_REPO_ROOT
for B in ${BANDS:-0 1 2}; do
  echo "=== AL_XCD_BAND=$B"
  SSLAM_EXTRA_HIPCC_FLAGS="-DAL_XCD_BAND=$B" python opencv-simpleslam_amd/build.py > /tmp/ab_build.log 2>&1 || { tail -5 /tmp/ab_build.log; continue; }
  python scripts/time_aliked.py 30 8 0; python scripts/time_aliked.py 30 1 0
  bash scripts/prof_aliked.sh 8 | grep "score_tail\|agg_pre\|resize_pad\|block1_rows\|block2_rows\|kernel time per frame"
  bash scripts/pmc_traffic_aliked.sh ab$B 8 | grep "score_tail\|agg_pre\|resize_pad\|block1_rows\|block2_rows\|sum per frame"
  python scripts/hash_aliked.py 2>&1 | tail -3
done
python opencv-simpleslam_amd/build.py > /dev/null 2>&1
