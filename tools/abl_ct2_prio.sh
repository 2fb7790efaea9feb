# convt2 experiments (ablation library): wave priorities by tap count, rolling B fragments, hot weight addresses
export RICK_HIP_LIB=rick_amd/lib/librick_hip_abl.so
for v in "0 0" "4 0" "5 0"; do
  set -- $v
  echo "== prio $1 rollb $2"
  RICK_CT2_PRIO=$1 RICK_CT2_ROLLB=$2 timeout 300 python tools/bench_conv.py dgrad 2>&1 | grep -E "^s2" | sed 's/fprop.*//'
done
