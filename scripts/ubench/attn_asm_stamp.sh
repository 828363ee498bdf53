# where a wave of the hand-scheduled attention kernel waits: s_memtime stamps around the tile barrier and the two fragment waits
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc scripts/ubench/attn_bench.hip -o /tmp/attn_b 2>/dev/null
for abl in "" novalu nomfma nodma; do
ATTN_ASM_ABL=$abl ATTN_ASM_STAMP=1 python3 opencv-simpleslam_amd/csrc/gen_lg_attention_asm.py > /tmp/a.s
/opt/rocm/lib/llvm/bin/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/a.s -o /tmp/a.o && /opt/rocm/lib/llvm/bin/ld.lld -shared /tmp/a.o -o /tmp/a.hsaco
echo -n "[$abl] "; ATTN_STAMP=1 ATTN_HSACO=/tmp/a.hsaco ATTN_PP=4 timeout -k 5 120 /tmp/attn_b 2048 8 1 5
done
