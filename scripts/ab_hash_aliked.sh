# bit-identity of an ALIKED kernel change on ONE box: ab_hash_aliked.sh "<flags A>" "<flags B>"  (SSLAM_EXTRA_HIPCC_FLAGS values); prints the diff of the hashes
export SSLAM_EXPERIMENT_BUILD=1     # build.py refuses SSLAM_EXTRA_HIPCC_FLAGS without it
cd $GRAFT_REPO_ROOT
i=0
for fl in "$@"; do
  SSLAM_EXTRA_HIPCC_FLAGS="$fl" python opencv-simpleslam_amd/build.py > /tmp/ab_build.log 2>&1 || { tail -5 /tmp/ab_build.log; exit 1; }
  python scripts/hash_aliked.py > /tmp/hash_$i.txt 2>&1
  i=$((i+1))
done
python opencv-simpleslam_amd/build.py > /dev/null 2>&1
if diff /tmp/hash_0.txt /tmp/hash_1.txt > /tmp/hash_diff.txt; then echo "IDENTICAL ($(wc -l < /tmp/hash_0.txt) cases)"; else echo "DIFFERENT"; cat /tmp/hash_diff.txt; fi
