# r06, VERDICT r05 item 8: WHAT has to run beside al_aggregate_kernel's unstable code shape for the events to appear.
# One build (the failing variant), scripts/diag_agg_rnorm.py with 1 / 2 / 3 extractor streams and with other work beside ONE
# extractor stream: a device-to-device copy stream (memory traffic only), a LightGlue matcher (MFMA / exp / LDS-DMA kernels).
export SSLAM_EXPERIMENT_BUILD=1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=${1:-400}; FL=${2:--DAL_AGG_FAST_SELU=2}
SSLAM_EXTRA_HIPCC_FLAGS="$FL" python opencv-simpleslam_amd/build.py > /tmp/diag_build.log 2>&1 || { tail -5 /tmp/diag_build.log; exit 1; }
CFGS=${3:-"1 none|2 none|3 none|1 copy|1 lightglue|2 lightglue"}
IFS='|' read -ra LIST <<< "$CFGS"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  T=$(echo $2 | tr ':' '_')
  timeout -k 10 600 python scripts/diag_agg_rnorm.py $R $1 $2 > gpurun_out/r06_agg_aggr_$1_$T.log 2>&1
  echo "$1 extractor stream(s) + $2: $(tail -1 gpurun_out/r06_agg_aggr_$1_$T.log)"
done
python opencv-simpleslam_amd/build.py > /dev/null 2>&1
