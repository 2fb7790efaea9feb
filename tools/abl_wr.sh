# wgrad second stage: tile kernel (runs of consecutive floats) against the element-per-thread kernel, same box
export RICK_HIP_LIB=rick_amd/lib/librick_hip_abl.so
for v in 0 1 0 1; do
  echo "== RICK_WR_TILE=$v"
  RICK_WR_TILE=$v B=${B:-8} timeout 300 python tools/bench_conv.py wgrad 2>&1 | grep -E "wgrad" | sed 's/| convT.*//'
done
