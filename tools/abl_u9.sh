#!/bin/bash
# unrolled 9-tap k-loop on short-K (Ci = 128 / 256) and stride-2 launches (needs make -C rick_amd/csrc abl)
export RICK_HIP_LIB=rick_amd/lib/librick_hip_abl.so
for cfg in "8 0" "4 0" "2 0" "8 1" "4 1"; do
  set -- $cfg
  echo "== RICK_U9_MINCHUNKS=$1 RICK_U9_S2=$2"
  RICK_U9_MINCHUNKS=$1 RICK_U9_S2=$2 python tools/bench_conv.py fprop 2>&1 | grep -v amdgpu | grep -E "@ 64|@128|@256|s2"
  RICK_U9_MINCHUNKS=$1 RICK_U9_S2=$2 B=8 python tools/bench_conv.py fprop 2>&1 | grep -v amdgpu | grep -E "s2" | sed 's/^/B8 /'
done
