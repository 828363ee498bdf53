# ping-pong (8-wave) vs 4-wave attention kernel on ONE device: attn_pp_ab.sh ["<extra flags>"]
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc $1 scripts/ubench/attn_bench.hip -o /tmp/attn_pp || exit 1
for rep in 1 2; do
  ATTN_PP=0 /tmp/attn_pp 2048 8 1 5
  ATTN_PP=1 /tmp/attn_pp 2048 8 1 5
done
ATTN_PP=1 /tmp/attn_pp 2048 2 1 5
ATTN_PP=1 /tmp/attn_pp 1900 8 1 5
