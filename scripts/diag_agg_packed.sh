# r06, VERDICT r05 item 8: the one-kernel bisect.  ONLY al_aggregate_kernel changes between the builds (the rest of the library
# is bit for bit the same code): the hardware exponential in its tail (AL_AGG_FAST_SELU=2, the switch that made r04 / r05's builds
# fail) with packed-fp32 instructions allowed in it (AL_AGG_PACKED=1: the SLP vectoriser's v_pk_mul_f32 / v_pk_fma_f32 shape) or
# not (the kernel's target attribute), and the product build; each under the two reproducers (three extractor streams; one
# extractor stream beside a LightGlue matcher on its ring GEMMs + HIP attention kernel).
export SSLAM_EXPERIMENT_BUILD=1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=${1:-400}
i=0
for fl in "-DAL_AGG_FAST_SELU=2 -DAL_AGG_PACKED=1" "-DAL_AGG_FAST_SELU=2" "-DAL_AGG_FAST_SELU=0"; do
  i=$((i + 1))
  SSLAM_EXTRA_HIPCC_FLAGS="$fl" python opencv-simpleslam_amd/build.py > /tmp/diag_build.log 2>&1 || { tail -5 /tmp/diag_build.log; continue; }
  for cfg in "3 none" "1 lightglue:ring,noasm"; do
    set -- $cfg
    T=$(echo $2 | tr ':,' '__')
    timeout -k 10 600 python scripts/diag_agg_rnorm.py $R $1 $2 > gpurun_out/r06_agg_packed_${i}_$1_$T.log 2>&1
    echo "[$fl] $1 extractor stream(s) + $2: $(tail -1 gpurun_out/r06_agg_packed_${i}_$1_$T.log)"
    grep "hypotheses reproducing" gpurun_out/r06_agg_packed_${i}_$1_$T.log | sed 's/ of level [^ ]*//; s/ ([0-9/]* px)//' | sort | uniq -c | sort -rn | head -3
  done
done
python opencv-simpleslam_amd/build.py > /dev/null 2>&1
