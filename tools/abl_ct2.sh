export RICK_HIP_LIB=rick_amd/lib/librick_hip_abl.so
for t in "" "16,8,1,1" "8,16,1,1" "32,4,1,1" "16,4,2,1" "13,9,1,1" "10,3,4,1"; do
  echo "== tile $t"; RICK_CT2_TILE=$t python tools/bench_conv.py dgrad 2>&1 | grep -E "s2" | head -3
done
