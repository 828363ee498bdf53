# Which aggressor the ISOLATED victim (al_aggregate_kernel alone in the failing shape, scripts/ubench/agg_victim.hip) needs:
#   agg_victim_aggressors.sh [rounds=300]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp SSLAM_ALLOW_RANDOM_WEIGHTS=1
R=${1:-300}
U=scripts/ubench
[ -f $U/libaggvictim_pk.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -shared -DAL_AGG_FAST_SELU=2 -DAL_AGG_PACKED=1 -o $U/libaggvictim_pk.so $U/agg_victim.hip 2>/dev/null
for spec in "none 1" "none 2" "copy 1" "lightglue 1" "lightglue:f32 1" "lightglue:layers1 1" "lightglue:ring 1" "lightglue:big 1" "lightglue:noasm 1" "lightglue:big,noasm 1" "lightglue:ring,noasm 1" \
            "lightglue:ring,noasm,layers1 1" "aliked:1 1" "aliked:2 1" \
            "synthetic:trans 1" "synthetic:mfma 1" "synthetic:pk 1" "synthetic:valu 1" "synthetic:lds 1" "synthetic:gather 1" "synthetic:store 1" "synthetic:scalar 1" "synthetic:ldsdma 1" \
            "synthetic:ldsdma:150:256:40 1" "synthetic:mfma:150:256:40 1" "synthetic:lds:150:256:40 1"; do
  set -- $spec
  timeout -k 10 200 python scripts/agg_victim_run.py $U/libaggvictim_pk.so $1 $R 40 $2 2 1 2>&1 | grep "rnorm words differing" | sed "s/^libaggvictim_pk.so //"
done
