# hand-scheduled attention kernel (opencv-simpleslam_amd/csrc/gen_lg_attention_asm.py): bitwise comparison against the 4-wave kernel, then A/B timing
cd $GRAFT_REPO_ROOT
set -e
python3 opencv-simpleslam_amd/csrc/gen_lg_attention_asm.py > /tmp/attn_asm.s
/opt/rocm/lib/llvm/bin/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/attn_asm.s -o /tmp/attn_asm.o
/opt/rocm/lib/llvm/bin/ld.lld -shared /tmp/attn_asm.o -o /tmp/attn_asm.hsaco
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc scripts/ubench/attn_bench.hip -o /tmp/attn_b 2>/dev/null
export ATTN_HSACO=/tmp/attn_asm.hsaco
set +e
echo "== N=256 1 pair self";   ATTN_PP=4 ATTN_CMP=1 timeout -k 5 60 /tmp/attn_b 256 1 1 1 | tail -6 || exit 1
echo "== N=2048 2 pairs self"; ATTN_PP=4 ATTN_CMP=1 timeout -k 5 60 /tmp/attn_b 2048 2 1 1 | tail -4 || exit 1
echo "== N=1900 ragged cross"; ATTN_PP=4 ATTN_CMP=1 ATTN_CROSS=1 ATTN_N1=1333 timeout -k 5 60 /tmp/attn_b 1900 2 1 1 | tail -4 || exit 1
echo "== N=200 N1=17 cross";   ATTN_PP=4 ATTN_CMP=1 ATTN_CROSS=1 ATTN_N1=17 timeout -k 5 60 /tmp/attn_b 200 1 1 1 | tail -4 || exit 1
echo "== N=64 self";           ATTN_PP=4 ATTN_CMP=1 timeout -k 5 60 /tmp/attn_b 64 3 1 1 | tail -4 || exit 1
for rep in 1 2; do
  echo -n "[asm pk] "; ATTN_ASM_PK=1 python3 opencv-simpleslam_amd/csrc/gen_lg_attention_asm.py > /tmp/p.s; /opt/rocm/lib/llvm/bin/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/p.s -o /tmp/p.o; /opt/rocm/lib/llvm/bin/ld.lld -shared /tmp/p.o -o /tmp/p.hsaco; ATTN_HSACO=/tmp/p.hsaco ATTN_PP=4 timeout -k 5 120 /tmp/attn_b 2048 8 1 5
  echo -n "[hs ] "; ATTN_PP=3 timeout -k 5 120 /tmp/attn_b 2048 8 1 5
  echo -n "[asm] "; ATTN_PP=4 timeout -k 5 120 /tmp/attn_b 2048 8 1 5
done
