# The evidence of a round's FINAL build, taken in one gpurun call so that every file names the same sources:
#   final_evidence.sh TAG GIT_HEAD        (GIT_HEAD: `git rev-parse --short HEAD` of the build container; the box has no .git)
# Writes gpurun_out/${TAG}_*: the GPU test log, the bench line, rocprofv3 kernel stats of the bench command, of one batched
# LightGlue forward (8 pairs), of one single-pair forward, of ALIKED at F = 8 and F = 1, the PMC traffic passes (LightGlue and ALIKED), the inter-kernel gaps of the single-frame ALIKED sequence, and
# ${TAG}_evidence_meta.json = {csrc_digest, git_head, files}: the digest of csrc/ + the header (build.py --digest) that
# tests/test_profiles_fresh.py compares with the tree once the files are copied into profiles/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r06}; HEAD=${2:-unknown}
export SSLAM_GIT_HEAD=$HEAD
mkdir -p gpurun_out
DIGEST=$(python opencv-simpleslam_amd/build.py --digest)
echo "== $TAG: csrc digest $DIGEST, head $HEAD"
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gputest.log 2>&1; tail -2 gpurun_out/${TAG}_gputest.log
timeout -k 10 900 python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err; tail -2 gpurun_out/${TAG}_bench_n1.err
bash scripts/prof_bench.sh $TAG 40 > gpurun_out/${TAG}_bench_concurrency.txt 2>&1
bash scripts/prof_lg_batch.sh 8 > gpurun_out/${TAG}_lightglue_batch8_kernels.txt 2>&1; cp gpurun_out/lgb_kernel_stats.csv gpurun_out/${TAG}_lightglue_batch8_kernel_stats.csv
bash scripts/prof_lg_batch.sh 1 > gpurun_out/${TAG}_lightglue_single_pair_kernels.txt 2>&1
bash scripts/prof_lg_batch.sh 2 > gpurun_out/${TAG}_lightglue_two_pair_kernels.txt 2>&1
bash scripts/prof_aliked.sh 8 > gpurun_out/${TAG}_aliked_F8_kernels.txt 2>&1; cp gpurun_out/aliked_kernel_stats_F8.csv gpurun_out/${TAG}_aliked_kernel_stats.csv
bash scripts/prof_aliked.sh 1 > gpurun_out/${TAG}_aliked_single_frame_kernels.txt 2>&1
bash scripts/pmc_traffic.sh $TAG 8 > gpurun_out/${TAG}_pmc_traffic.txt 2>&1
bash scripts/pmc_traffic_aliked.sh $TAG 8 > gpurun_out/${TAG}_pmc_traffic_aliked.txt 2>&1
bash scripts/gaps_aliked.sh 1 > gpurun_out/${TAG}_aliked_graph_gaps.txt 2>&1
rm -f gpurun_out/pmc_FETCH_SIZE.* gpurun_out/pmc_WRITE_SIZE.* gpurun_out/prof_gap.log
# the drop-in loops once more on this box, duck types and cv2's classes back to back (bench.py ran them minutes apart), and the ring's
# extractor as a cached graph against plain launches
{ for G in 0 1; do SSLAM_RING_GRAPHS=$G SSLAM_ALLOW_RANDOM_WEIGHTS=1 python scripts/dropin_bench.py 96 2>/dev/null | python -c "
import json, sys
d = json.load(sys.stdin)
print('ring graphs $G, duck types : value', d['value'], 'slam_loop', d['slam_loop']['value'], 'frame_loop', d['frame_loop']['value'], 'extractor / matcher / filter ms', d['feature_extractor_ms'], d['feature_matcher_ms'], d['filter_matches_ransac_ms'])"
  SSLAM_RING_GRAPHS=$G SSLAM_ALLOW_RANDOM_WEIGHTS=1 python scripts/dropin_bench_cv2.py 96 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.load(sys.stdin)
print('ring graphs $G, cv2 classes: value', d['value'], 'slam_loop', d['slam_loop']['value'], 'extractor / matcher / filter ms', d['feature_extractor_ms'], d['feature_matcher_ms'], d['filter_matches_ransac_ms'])"
done; } > gpurun_out/${TAG}_dropin_back_to_back.txt 2>&1
python - $TAG $DIGEST $HEAD <<'PY'
import glob, json, os, sys
tag, digest, head = sys.argv[1:4]
files = sorted(os.path.basename(f) for f in glob.glob(f"gpurun_out/{tag}_*") if not f.endswith("_evidence_meta.json"))
json.dump({"tag": tag, "csrc_digest": digest, "git_head": head, "files": files,
           "what": "scripts/final_evidence.sh: every file listed was produced by ONE gpurun call on the sources this digest names"},
          open(f"gpurun_out/{tag}_evidence_meta.json", "w"), indent=1)
print("evidence files:", len(files))
PY
