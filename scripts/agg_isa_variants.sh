# Builds patched code objects of al_aggregate_kernel in the failing shape (scripts/agg_isa_patch.py) into scripts/ubench/isa/ - runs where
# hipcc is (the build container or the box); then  agg_isa_run.sh  loops each beside the LightGlue trigger.
#   agg_isa_variants.sh MODE | NAME@MODE ...       (NAME: the code object's file name, for modes with blanks in them)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
D=scripts/ubench/isa; mkdir -p $D
L=/opt/rocm/lib/llvm/bin
[ -f $D/vic.s ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAL_AGG_FAST_SELU=2 -DAL_AGG_PACKED=1 --cuda-device-only -S -o $D/vic.s scripts/ubench/agg_victim.hip 2>/dev/null
for a in "$@"; do
  case "$a" in *@*) n="${a%%@*}"; m="${a#*@}";; *) m="$a"; n=$(echo "$a" | tr ':,' '__');; esac
  python scripts/agg_isa_patch.py $D/vic.s "$D/$n.s" "$m" | grep -v "^pk_split" ; $L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$D/$n.s" -o "$D/$n.o" && $L/ld.lld -shared "$D/$n.o" -o "$D/$n.co" || echo "FAILED $n"
  rm -f "$D/$n.o"
done
