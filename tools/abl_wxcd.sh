# wgrad block -> XCD placement for 1 / 2 / 4 splits (RICK_CONV_DEBUG=16: the round-robin placement), same box
export RICK_HIP_LIB=rick_amd/lib/librick_hip_abl.so
for v in 16 0 16 0; do
  echo "== RICK_CONV_DEBUG=$v"
  for b in 4 8; do RICK_CONV_DEBUG=$v B=$b timeout 300 python tools/bench_conv.py wgrad 2>&1 | grep -E "wgrad" | grep -E "@ 64|@128|@256|@ 32" | sed "s/^/B=$b /;s/| convT.*//"; done
done
