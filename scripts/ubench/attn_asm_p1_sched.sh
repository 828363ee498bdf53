# r04: schedule knobs of the P-one-plane assembly kernel (exp2 pairs taken in the ODD half, which has 8 MFMA slots there)
cd $GRAFT_REPO_ROOT
set -e
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc scripts/ubench/attn_bench.hip -o /tmp/attn_b 2>/dev/null
set +e
for n in 4 3 2 1 0 5; do
  ATTN_ASM_P1=1 ATTN_ASM_NEXP_ODD=$n python3 opencv-simpleslam_amd/csrc/gen_lg_attention_asm.py > /tmp/q.s
  /opt/rocm/lib/llvm/bin/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/q.s -o /tmp/q.o && /opt/rocm/lib/llvm/bin/ld.lld -shared /tmp/q.o -o /tmp/q.hsaco
  for rep in 1 2; do echo -n "[P1 NEXP_ODD=$n] "; ATTN_HSACO=/tmp/q.hsaco ATTN_PP=4 timeout -k 5 120 /tmp/attn_b 2048 8 1 5; done
done
