# convt2: the planner's sub-block tail (default) against whole-tile blocks only (RICK_CT2_SUBQ=1), same box
export RICK_HIP_LIB=rick_amd/lib/librick_hip_abl.so
for q in 1 auto 1 auto; do
  echo "== subq $q"
  if [ $q = auto ]; then unset RICK_CT2_SUBQ; else export RICK_CT2_SUBQ=$q; fi
  timeout 300 python tools/ct2_rounds.py 2>&1 | grep rounds | grep -E "B=(2|4|8):"
done
