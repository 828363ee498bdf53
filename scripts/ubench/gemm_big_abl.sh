# ablation of the big GEMM main loop: which part sets the time (run on the GPU box)
# GEMM_ABL bits: 1 no MFMA, 2 no DMA in the loop, 4 no fragment reads, 8 no stores
cd $GRAFT_REPO_ROOT
for abl in ${ABLS:-0 15 13 14 11 12 9 10}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGEMM_ABL=$abl -I opencv-simpleslam_amd/csrc scripts/ubench/gemm_big_bench.hip -o /tmp/gemm_big_$abl 2>/dev/null
  echo "== GEMM_ABL=$abl"
  /tmp/gemm_big_$abl quick | grep "128x256 w8 v[01]\|128x128 w4"
done
