export SSLAM_EXPERIMENT_BUILD=1     # build.py refuses SSLAM_EXTRA_HIPCC_FLAGS without it
set -e
cd $GRAFT_REPO_ROOT
for flags in "" "-DSSLAM_DBG_NOMFMA=1" "-DSSLAM_DBG_NOEPI=1" "-DSSLAM_DBG_NOMFMA=1 -DSSLAM_DBG_NOEPI=1"; do
  SSLAM_EXTRA_HIPCC_FLAGS="$flags" python opencv-simpleslam_amd/build.py > /dev/null 2>&1
  echo "== flags: [$flags]"
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/px && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px -- python $GRAFT_REPO_ROOT/scripts/time_lightglue.py 2048 5 > /dev/null 2>&1
  python - <<'PY'
import csv,glob
f=sorted(glob.glob('/tmp/px/*/*_kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if 'linear_h' in r['Name']: print('   ', r['Name'][40:75], r['AverageNs'][:7])
PY
  cd $GRAFT_REPO_ROOT
done
SSLAM_EXTRA_HIPCC_FLAGS="" python opencv-simpleslam_amd/build.py --force > /dev/null   # never leave an experiment build behind
