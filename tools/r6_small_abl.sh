#!/bin/bash
# round 6, item 3: where the time of a small split-K conv launch goes (ablation build: RICK_CONV_DEBUG 1 = no MFMA, 2 = no patch prefetch, 4 = no weight loads)
out=gpurun_out/small_abl; mkdir -p $out
export RICK_HIP_LIB=rick_amd/lib/librick_hip_abl.so RICK_TUNE=2=0
for d in 0 4 7; do
  RICK_CONV_DEBUG=$d python tools/bench_small.py 2>/dev/null | grep "conv s" | sed "s/^/debug=$d /" | cut -c1-150 > $out/d$d.txt
done
paste -d'\n' $out/d0.txt $out/d4.txt $out/d7.txt
