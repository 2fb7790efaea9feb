#!/bin/bash
# unrolled 9-tap k-loop on split-K launches (needs make -C rick_amd/csrc abl)
export RICK_HIP_LIB=rick_amd/lib/librick_hip_abl.so
for sp in 0 8 4 2; do
  echo "== RICK_U9_SPLIT=$sp"
  RICK_U9_SPLIT=$sp python tools/bench_conv.py fprop dgrad 2>&1 | grep -v amdgpu | grep -E "@ 16|@ 32|@ 64|s2"
done
