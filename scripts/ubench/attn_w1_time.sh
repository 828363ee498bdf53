cd $GRAFT_REPO_ROOT
i=0
for f in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc $f scripts/ubench/attn_bench.hip -o /tmp/attn_t$i 2>/dev/null || echo "build failed: $f"
  i=$((i+1))
done
ATTN_PP=0 /tmp/attn_t0 2048 8 1 5
i=0
for f in "$@"; do
  echo "== [$f]"; ATTN_PP=2 ATTN_CMP=1 /tmp/attn_t$i 2048 2 1 1 | tail -1
  ATTN_PP=2 /tmp/attn_t$i 2048 8 1 5; ATTN_PP=2 ATTN_ZERO=1 /tmp/attn_t$i 2048 8 1 5
  i=$((i+1))
done
