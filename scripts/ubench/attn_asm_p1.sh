# r04: what the P-one-plane instruction stream would take (profiles/r04_split_study.md): the assembly kernel built with
# ATTN_ASM_P1=1 (20 MFMAs per sub-step, no P.lo arithmetic; results are NOT valid) against the product kernel, same box
cd $GRAFT_REPO_ROOT
set -e
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc scripts/ubench/attn_bench.hip -o /tmp/attn_b 2>/dev/null
for tag in 0 1; do
  ATTN_ASM_P1=$tag python3 opencv-simpleslam_amd/csrc/gen_lg_attention_asm.py > /tmp/p$tag.s
  /opt/rocm/lib/llvm/bin/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/p$tag.s -o /tmp/p$tag.o
  /opt/rocm/lib/llvm/bin/ld.lld -shared /tmp/p$tag.o -o /tmp/p$tag.hsaco
done
set +e
for rep in 1 2 3; do
  echo -n "[asm product ] "; ATTN_HSACO=/tmp/p0.hsaco ATTN_PP=4 timeout -k 5 120 /tmp/attn_b 2048 8 1 5
  echo -n "[asm P1 20mfma] "; ATTN_HSACO=/tmp/p1.hsaco ATTN_PP=4 timeout -k 5 120 /tmp/attn_b 2048 8 1 5
done
