# The single-instruction probe (scripts/ubench/pk_probe.hip) beside an aggressor: mismatching results per mode.
#   pk_probe_run.sh [rounds=200] [launches per round=10]        (BESIDE="<aggressor spec> ...", MODES="0 1 ...")
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp SSLAM_ALLOW_RANDOM_WEIGHTS=1
R=${1:-200}; I=${2:-10}
U=scripts/ubench
[ -f $U/libpkprobe.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -shared -o $U/libpkprobe.so $U/pk_probe.hip 2>/dev/null
for B in ${BESIDE:-synthetic:mfma16 lightglue:big,noasm}; do
for m in ${MODES:-$(seq 0 97)}; do
  T=$(python -c "
import ctypes; V = ctypes.CDLL('$U/libpkprobe.so'); V.victim_mode_text.restype = ctypes.c_char_p; print((V.victim_mode_text($m & 255) or b'?').decode() + (' + loads' if $m >> 8 else ''))")
  timeout -k 10 300 python scripts/agg_victim_run.py $U/libpkprobe.so $B $R $I 1 $m 1 2>&1 | grep "words differing\|Error\|assert" | sed "s/rnorm words differing/LOW-half mismatches/; s/s8 words differing/HIGH-half mismatches/; s/(runs of 16: [0-9.]*), //; s/libpkprobe.so beside //; s/: [0-9]* rounds.*launches in/:/" | sed "s|^|mode $m  $T  beside |"
done; done
