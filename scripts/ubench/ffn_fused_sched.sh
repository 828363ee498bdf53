# fused FFN ubench: scheduling variants x ablations (run on the GPU box)
cd $GRAFT_REPO_ROOT
for sch in ${SCHEDS:-0 1 2}; do
for abl in ${ABLS:-0 1}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFFN_ABL=$abl -DFFN_SCHED=$sch ${FFN_FLAGS} -I opencv-simpleslam_amd/csrc scripts/ubench/ffn_fused_bench.hip -o /tmp/ffn_fused_${sch}_$abl 2>/dev/null || { echo "compile failed ($sch $abl)"; continue; }
  echo "== FFN_SCHED=$sch FFN_ABL=$abl ${FFN_FLAGS}"
  timeout -k 5 120 /tmp/ffn_fused_${sch}_$abl 32768 20
done
done
