# bit-identity of a LightGlue kernel change on ONE box: ab_hash_lightglue.sh "<flags A>" "<flags B>"  (SSLAM_EXTRA_HIPCC_FLAGS values); prints the diff of the hashes
export SSLAM_EXPERIMENT_BUILD=1     # build.py refuses SSLAM_EXTRA_HIPCC_FLAGS without it
cd $GRAFT_REPO_ROOT
i=0
for fl in "$@"; do
  SSLAM_EXTRA_HIPCC_FLAGS="$fl" python opencv-simpleslam_amd/build.py > /tmp/ab_build.log 2>&1 || { tail -5 /tmp/ab_build.log; exit 1; }
  python scripts/hash_lightglue.py > /tmp/hash_lg_$i.txt 2>&1
  python scripts/time_lightglue_batch.py 2048 8 10 | tail -1
  i=$((i+1))
done
python opencv-simpleslam_amd/build.py > /dev/null 2>&1
cat /tmp/hash_lg_0.txt
if diff /tmp/hash_lg_0.txt /tmp/hash_lg_1.txt > /tmp/hash_lg_diff.txt; then echo "IDENTICAL"; else echo "DIFFERENT"; cat /tmp/hash_lg_diff.txt; fi
