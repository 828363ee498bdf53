# which build switch makes ALIKED non-deterministic under concurrency: ab_stress_aliked.sh <repeats> "<flags A>" "<flags B>" ...
# (per variant: the library rebuilt with the flags, scripts/stress_aliked_repeat.py <repeats> 3 2 - 3 extractor instances x 2 frames x 4
#  un-synchronised calls per repeat, every stage buffer hashed against the first run; details in gpurun_out/ab_stress_<n>.log)
export SSLAM_EXPERIMENT_BUILD=1     # build.py refuses SSLAM_EXTRA_HIPCC_FLAGS without it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$1; shift
i=0
for fl in "$@"; do
  i=$((i + 1))
  echo "=== flags: $fl"
  SSLAM_EXTRA_HIPCC_FLAGS="$fl" python opencv-simpleslam_amd/build.py > /tmp/ab_build.log 2>&1 || { tail -5 /tmp/ab_build.log; continue; }
  timeout -k 10 900 python scripts/stress_aliked_repeat.py $R 3 2 > gpurun_out/ab_stress_$i.log 2>&1
  echo "events (a repeat whose stage buffers differ): $(grep -c 'stage buffers differ' gpurun_out/ab_stress_$i.log)"
  grep 'differ' gpurun_out/ab_stress_$i.log | head -6
  tail -1 gpurun_out/ab_stress_$i.log
done
python opencv-simpleslam_amd/build.py > /dev/null 2>&1
