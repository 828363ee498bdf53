# what the wrong 1/||F|| values of al_aggregate_kernel's unstable code shape ARE (scripts/diag_agg_rnorm.py),
# and whether the events need kernels of different streams to overlap (AMD_SERIALIZE_KERNEL=3: the runtime waits in front of and
# behind every launch).     diag_agg_rnorm.sh [flags="-DAL_AGG_FAST_SELU=2"] [repeats=120]
export SSLAM_EXPERIMENT_BUILD=1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
FL=${1:--DAL_AGG_FAST_SELU=2}; R=${2:-120}
SSLAM_EXTRA_HIPCC_FLAGS="$FL" python opencv-simpleslam_amd/build.py > /tmp/diag_build.log 2>&1 || { tail -5 /tmp/diag_build.log; exit 1; }
echo "=== $FL, $R repeats, streams free to overlap"
timeout -k 10 900 python scripts/diag_agg_rnorm.py $R > gpurun_out/r06_agg_diag.log 2>&1; tail -3 gpurun_out/r06_agg_diag.log
echo "=== the same build, AMD_SERIALIZE_KERNEL=3"
AMD_SERIALIZE_KERNEL=3 timeout -k 10 600 python scripts/diag_agg_rnorm.py $((R / 2)) > gpurun_out/r06_agg_diag_serialized.log 2>&1; tail -1 gpurun_out/r06_agg_diag_serialized.log
python opencv-simpleslam_amd/build.py > /dev/null 2>&1
