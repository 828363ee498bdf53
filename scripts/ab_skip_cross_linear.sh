# Upper bound of what emitting the cross block's qk / v planes from the self block's FFN could buy IN THE PIPELINE: the cross
# projection launch left out altogether (results wrong - timing only) against the product, same box, alternating.
# Needs the guarded lines described at the end of profiles/r06_linear_emission_bound.md around the cross projection launch in lg_layer_h
# (they are not kept in the product source).
export SSLAM_EXPERIMENT_BUILD=1
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for fl in "" "-DSSLAM_DBG_SKIP_CROSS_LINEAR=1"; do
    SSLAM_EXTRA_HIPCC_FLAGS="$fl" python opencv-simpleslam_amd/build.py > /tmp/ab_build.log 2>&1 || { tail -5 /tmp/ab_build.log; continue; }
    v=$(python bench.py --no-extras --no-cpu-baseline --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['lightglue_batch_ms_isolated'])")
    echo "flags [$fl]: frames/s, ms per round, isolated 8-pair forward ms: $v"
  done
done
unset SSLAM_EXTRA_HIPCC_FLAGS
python opencv-simpleslam_amd/build.py > /dev/null 2>&1
