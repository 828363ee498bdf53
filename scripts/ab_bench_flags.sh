# A/B of build flags on the BENCH itself (the pipeline, where kernels of four streams share the chip): ab_bench_flags.sh "<flags A>" "<flags B>" ...
export SSLAM_EXPERIMENT_BUILD=1     # build.py refuses SSLAM_EXTRA_HIPCC_FLAGS without it
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for fl in "$@"; do
  SSLAM_EXTRA_HIPCC_FLAGS="$fl" python opencv-simpleslam_amd/build.py > /tmp/ab_build.log 2>&1 || { tail -5 /tmp/ab_build.log; continue; }
  echo -n "flags [$fl]: "
  python bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['step_ms']['p50'])"
done
done
python opencv-simpleslam_amd/build.py > /dev/null 2>&1
