# attn_mix.sh: "<PP> <flags>" entries, e.g. "0 -DATTN_WAVES_PER_SIMD=1"
cd $GRAFT_REPO_ROOT
i=0
for e in "$@"; do
  f="${e#* }"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc $f scripts/ubench/attn_bench.hip -o /tmp/attn_$i 2>/dev/null || echo "build failed: $f"
  i=$((i+1))
done
for rep in 1 2; do
  i=0
  for e in "$@"; do pp="${e%% *}"; echo -n "[$e] "; ATTN_PP=$pp /tmp/attn_$i 2048 8 1 5; i=$((i+1)); done
done
