# ablations of the ping-pong attention kernel: attn_pp_abl.sh "<flags A>" "<flags B>" ...
cd $GRAFT_REPO_ROOT
i=0
for f in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc $f scripts/ubench/attn_bench.hip -o /tmp/attn_$i 2>/dev/null || echo "build failed: $f"
  i=$((i+1))
done
for rep in 1 2; do
  i=0
  for f in "$@"; do echo -n "[$f] "; ATTN_PP=1 /tmp/attn_$i 2048 8 1 5; i=$((i+1)); done
done
