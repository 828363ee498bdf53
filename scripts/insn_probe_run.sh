# Every modifier-bearing instruction form of the product's ISA as a victim (scripts/ubench/insn_probe.hip): threads whose result hash
# differs from the solitary launch's, beside the strongest triggers.
#   insn_probe_run.sh [rounds=200] [launches per round=10]        (BESIDE="<aggressor spec> ...", MODES="0 1 ...")
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp SSLAM_ALLOW_RANDOM_WEIGHTS=1
R=${1:-200}; I=${2:-10}
U=scripts/ubench
[ -f $U/libinsnprobe.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -shared -o $U/libinsnprobe.so $U/insn_probe.hip 2>/dev/null
for B in ${BESIDE:-synthetic:mfma_16x16x32_f16 lightglue:big,noasm}; do
for m in ${MODES:-$(seq 0 28)}; do
  T=$(python -c "
import ctypes; V = ctypes.CDLL('$U/libinsnprobe.so'); V.victim_mode_text.restype = ctypes.c_char_p; print((V.victim_mode_text($m) or b'?').decode())")
  timeout -k 10 300 python scripts/agg_victim_run.py $U/libinsnprobe.so $B $R $I 1 $m 1 2>&1 | grep "words differing\|Error\|assert" | sed "s/rnorm words differing/threads whose hash differs/; s/, s8 words differing 0//; s/(runs of 16: [0-9.]*)//; s/libinsnprobe.so beside //; s/: [0-9]* rounds.*launches in/:/" | sed "s|^|$m  $T  beside |"
done; done
