# quick sweep of the pipeline shape on one box: frames per round / matcher streams / pairs per batch / frames per extractor call
cd $GRAFT_REPO_ROOT
run() { echo -n "$* -> "; env "$@" python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['step_ms']['p50'])"; }
run SSLAM_BENCH_FRAMES=24
run SSLAM_BENCH_FRAMES=32
run SSLAM_BENCH_FRAMES=48 SSLAM_BENCH_EF=16
run SSLAM_BENCH_FRAMES=24 SSLAM_BENCH_NM=4
run SSLAM_BENCH_FRAMES=24 SSLAM_BENCH_NM=2
run SSLAM_BENCH_FRAMES=24 SSLAM_BENCH_EF=12
run SSLAM_BENCH_FRAMES=24 SSLAM_BENCH_EF=6
run SSLAM_BENCH_FRAMES=24 SSLAM_BENCH_PAIRS=12 SSLAM_BENCH_NM=2
run SSLAM_BENCH_FRAMES=24
