# where the tile DMA pieces sit in the EVEN half of the hand-scheduled attention kernel (first MFMA slot that carries one)
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc scripts/ubench/attn_bench.hip -o /tmp/attn_b 2>/dev/null
for rep in 1 2; do
for f in 1 4 8; do
  ATTN_ASM_DMA_FIRST=$f python3 opencv-simpleslam_amd/csrc/gen_lg_attention_asm.py > /tmp/a.s
  /opt/rocm/lib/llvm/bin/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/a.s -o /tmp/a.o && /opt/rocm/lib/llvm/bin/ld.lld -shared /tmp/a.o -o /tmp/a.hsaco || continue
  if [ $rep = 1 ]; then echo -n "[first slot $f] "; ATTN_HSACO=/tmp/a.hsaco ATTN_PP=4 ATTN_CMP=1 ATTN_CROSS=1 ATTN_N1=1333 timeout -k 5 60 /tmp/attn_b 1900 2 1 1 | grep bitwise; fi
  echo -n "[first slot $f] "; ATTN_HSACO=/tmp/a.hsaco ATTN_PP=4 timeout -k 5 120 /tmp/attn_b 2048 8 1 5
done
done
