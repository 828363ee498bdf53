# fused FFN ubench with its ablations (run on the GPU box): ffn_fused_abl.sh [abl list]
cd $GRAFT_REPO_ROOT
for abl in ${ABLS:-0 1 2 4 8 3 7}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFFN_ABL=$abl ${FFN_FLAGS} -I opencv-simpleslam_amd/csrc scripts/ubench/ffn_fused_bench.hip -o /tmp/ffn_fused_$abl 2>/dev/null || { echo "compile failed ($abl)"; continue; }
  echo "== FFN_ABL=$abl ${FFN_FLAGS}"
  timeout -k 5 120 /tmp/ffn_fused_$abl 32768 20
done
