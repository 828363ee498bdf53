# A/B of two builds of libsslam_hip.so on ONE GPU box (box-to-box spread is ~5 %, larger than most
# kernel-level effects): ab_lib.sh "<hipcc flags A>" "<hipcc flags B>" [pairs=8]
# Builds each variant in place (content-keyed objects), profiles the batched LightGlue forward with
# rocprofv3 and prints the per-kernel table of each; the default build is restored at the end.
export SSLAM_EXPERIMENT_BUILD=1     # build.py refuses SSLAM_EXTRA_HIPCC_FLAGS without it
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B=${3:-8}
i=0
for f in "$1" "$2"; do
  SSLAM_EXTRA_HIPCC_FLAGS="$f" python opencv-simpleslam_amd/build.py > /dev/null 2>&1 || { echo "build failed: $f"; exit 1; }
  echo "=== variant $i: [$f]"
  bash scripts/prof_lg_batch.sh $B | head -9 | cut -c1-140
  bash scripts/prof_lg_batch.sh $B | tail -1
  i=$((i+1))
done
python opencv-simpleslam_amd/build.py > /dev/null 2>&1
