# one-wave-per-SIMD attention kernel: compare against the 4-wave kernel, then time both
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc $1 scripts/ubench/attn_bench.hip -o /tmp/attn_w1 2>/dev/null || exit 1
ATTN_PP=2 ATTN_CMP=1 /tmp/attn_w1 256 1 1 1
ATTN_PP=2 ATTN_CMP=1 /tmp/attn_w1 2048 2 1 1 | tail -8
for rep in 1 2; do
  ATTN_PP=0 /tmp/attn_w1 2048 8 1 5
  ATTN_PP=2 /tmp/attn_w1 2048 8 1 5
done
ATTN_PP=2 ATTN_ZERO=1 /tmp/attn_w1 2048 8 1 5
