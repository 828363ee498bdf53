# al_aggregate_kernel ALONE as the victim of the co-residency fault (scripts/ubench/agg_victim.hip, scripts/agg_victim_run.py):
#   agg_victim.sh [rounds=300]
# builds the failing shape (packed fp32 allowed, v_exp_f32 tail) and the product's shape and loops each beside the aggressors
# that profiles/r06_aggregate_rnorm_diagnosis.md ranks, then the failing shape with no aggressor and with extractor-free variants.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp SSLAM_ALLOW_RANDOM_WEIGHTS=1
R=${1:-300}
U=scripts/ubench
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -shared"
[ -f $U/libaggvictim_pk.so ] || $B -DAL_AGG_FAST_SELU=2 -DAL_AGG_PACKED=1 -o $U/libaggvictim_pk.so $U/agg_victim.hip 2>/dev/null
[ -f $U/libaggvictim_product.so ] || $B -o $U/libaggvictim_product.so $U/agg_victim.hip 2>/dev/null
for spec in "pk lightglue:ring,noasm 1" "pk none 1" "pk none 3" "product lightglue:ring,noasm 1" "product none 3"; do
  set -- $spec
  echo "== victim $1 beside $2, $3 stream(s)"
  timeout -k 10 300 python scripts/agg_victim_run.py $U/libaggvictim_$1.so $2 $R 40 $3 2 1 2>&1 | grep -v "^  round" | tail -20
done
