# A/B of the 4-wave FFN kernel's build knobs on one box: kernel time per 8-pair launch under the tracer
export SSLAM_EXPERIMENT_BUILD=1     # build.py refuses SSLAM_EXTRA_HIPCC_FLAGS without it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for fl in "" "-DFFN4_D2=8" "-DFFN4_D1=4" "-DFFN4_D1=4 -DFFN4_D2=8"; do
  SSLAM_EXTRA_HIPCC_FLAGS="$fl" python opencv-simpleslam_amd/build.py > /tmp/ab_build.log 2>&1 || { tail -5 /tmp/ab_build.log; continue; }
  for m in 2 6; do
    SSLAM_BIG_GEMM=$m rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf -- python scripts/time_lightglue_batch.py 2048 8 4 > /dev/null 2>&1
    echo -n "flags [$fl] mode $m: "; find gpurun_out/pf -name "*kernel_stats.csv" -exec grep "ffn_fused" {} \; | awk -F, '{print $1, "avg_ns", $(NF-4)}' | cut -c1-140; rm -rf gpurun_out/pf
  done
done
python opencv-simpleslam_amd/build.py > /dev/null 2>&1
