# Which ONE instruction of another queue's waves makes the probe instruction (scripts/ubench/pk_probe.hip, mode 6 = v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]) fail:
#   pk_probe_aggressors.sh [rounds=200] [kinds...]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp SSLAM_ALLOW_RANDOM_WEIGHTS=1
R=${1:-200}; shift
U=scripts/ubench
KINDS="$@"; [ -n "$KINDS" ] || KINDS="insn_all mixlo mixhi sdwa cvtpk perm permswap bitop3 mov64 max3 pkmul_hi10 pkmul_01 pkfma_hi101 mfma16 cvtf16 fmamk bfi cmpabs lshladd64 exp pkmul cvt_f16 shl64 trans mfma pk valu lds gather store scalar ldsdma"
for k in $KINDS; do
  timeout -k 10 300 python scripts/agg_victim_run.py $U/libpkprobe.so synthetic:$k $R 10 1 ${MODE:-6} 1 2>&1 | grep "words differing\|Error\|assert" | sed "s/rnorm words differing/LOW-half mismatches/; s/s8 words differing/HIGH-half mismatches/; s/(runs of 16: [0-9.]*), //; s/libpkprobe.so beside //"
done
