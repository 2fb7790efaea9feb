# igemm block order: co tile fastest (default) against position tile fastest (RICK_CONV_DEBUG=32), same box
export RICK_HIP_LIB=rick_amd/lib/librick_hip_abl.so
for v in 32 0 32 0; do
  echo "== RICK_CONV_DEBUG=$v"
  for b in 4 8; do RICK_CONV_DEBUG=$v B=$b timeout 300 python tools/bench_conv.py fprop dgrad 2>&1 | grep -E "fprop" | grep -E "@ 64|@128|@256|@ 32|@ 16" | sed "s/^/B=$b /;s/| convT.*//"; done
done
