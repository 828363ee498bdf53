# Loops every code object of scripts/ubench/isa/ (or the ones named) beside the LightGlue trigger: events per 12 000 launches.
#   agg_isa_run.sh [rounds=300] [name ...]        (VICTIM_REF_PATCHED=1, SHOW=n: the first n events of each in full)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp SSLAM_ALLOW_RANDOM_WEIGHTS=1
R=${1:-300}; shift
U=scripts/ubench
[ -f $U/libaggvictim_pk.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -shared -DAL_AGG_FAST_SELU=2 -DAL_AGG_PACKED=1 -o $U/libaggvictim_pk.so $U/agg_victim.hip 2>/dev/null
NAMES="$@"; [ -n "$NAMES" ] || NAMES=$(ls $U/isa/*.co | xargs -n1 basename | sed 's/\.co$//')
for n in $NAMES; do
  timeout -k 10 200 python scripts/agg_victim_run.py $U/libaggvictim_pk.so:$U/isa/$n.co ${BESIDE:-lightglue:big,noasm} $R 40 1 2 1 2>&1 | grep "rnorm words differing\|Error\|assert\|stream 0" | awk -v show=${SHOW:-0} '/stream 0/{ if (++k <= show) print; next } {k=0; print}' | sed "s/300 rounds x 40 launches x 1 streams x 2 frames = //"
done
