# half-step attention kernel: correctness against the 4-wave kernel, then A/B timing on ONE device
cd $GRAFT_REPO_ROOT
i=0
for f in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc $f scripts/ubench/attn_bench.hip -o /tmp/attn_$i 2>/dev/null || echo "build failed: $f"
  i=$((i+1))
done
ATTN_PP=3 ATTN_CMP=1 timeout -k 5 120 /tmp/attn_0 2048 2 1 1 | tail -4
ATTN_PP=3 ATTN_CMP=1 timeout -k 5 120 /tmp/attn_0 1900 1 1 1 | tail -3
ATTN_PP=3 ATTN_CMP=1 timeout -k 5 120 /tmp/attn_0 200 1 1 1 | tail -3
for rep in 1 2; do
  echo -n "[p4 baseline] "; ATTN_PP=0 timeout -k 5 120 /tmp/attn_0 2048 8 1 5
  i=0
  for f in "$@"; do echo -n "[hs $f] "; ATTN_PP=3 timeout -k 5 120 /tmp/attn_$i 2048 8 1 5; i=$((i+1)); done
done
