# random vs all-zero operands (same instruction stream): how much of the launch time is the chip holding its clock down
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc scripts/ubench/attn_bench.hip -o /tmp/attn_z 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc -DATTN_ABL=14 scripts/ubench/attn_bench.hip -o /tmp/attn_z14 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc -DATTN_ABL=1 scripts/ubench/attn_bench.hip -o /tmp/attn_z1 2>/dev/null
for rep in 1 2; do
  for z in 0 1; do
    echo -n "zero=$z full   "; ATTN_ZERO=$z /tmp/attn_z 2048 8 1 5
    echo -n "zero=$z mfma   "; ATTN_ZERO=$z /tmp/attn_z14 2048 8 1 5
    echo -n "zero=$z nomfma "; ATTN_ZERO=$z /tmp/attn_z1 2048 8 1 5
  done
done
