# A/B of attention schedule variants on ONE device (run on the GPU box): attn_ab.sh "<flags A>" "<flags B>" ...
cd $GRAFT_REPO_ROOT
i=0
for f in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc $f scripts/ubench/attn_bench.hip -o /tmp/attn_$i 2>/dev/null || echo "build failed: $f"
  i=$((i+1))
done
for rep in 1 2; do
  i=0
  for f in "$@"; do echo -n "[$f] "; /tmp/attn_$i 2048 8 1 5; i=$((i+1)); done
done
