cd $GRAFT_REPO_ROOT
for k in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc -DW1_PADN=3 -DW1_SKIP=$k scripts/ubench/attn_bench.hip -o /tmp/attn_a 2>/dev/null || echo "build failed: $k"
  echo -n "skip=$k "; ATTN_PP=2 ATTN_ZERO=1 /tmp/attn_a 2048 8 1 3 | cut -c1-110
done
