#!/bin/bash
# round 6: the FIR's 16 x 16 x 32-channel tile against the 8 x 8 x 64 one — parity, micro-benchmark, in-situ A/B
out=gpurun_out/ufd16; mkdir -p $out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_split.py -q -m gpu -x -k "upfirdn or tile16 or resblock or blur or dblock" > $out/tests.txt 2>&1; tail -3 $out/tests.txt
python -m pytest tests/test_gpu_models.py -q -m gpu -x -k "one_node or full_256" >> $out/tests.txt 2>&1; tail -2 $out/tests.txt
python tools/bench_elem.py 2>&1 | grep "ch @" | sed 's/^/tile16 /' > $out/elem16.txt
RICK_TUNE=3=0 python tools/bench_elem.py 2>&1 | grep "ch @" | sed 's/^/tile8  /' > $out/elem8.txt
paste -d'\n' $out/elem16.txt $out/elem8.txt | awk '{print $1, $2, $3, $4, $5, $6, $7; for(i=8;i<=NF;i++) if ($i ~ /blur/) printf "   %s %s %s", $i, $(i+1), $(i+2); print ""}'
tools/ab_env2.sh RICK_TUNE=3=0 2 2>&1 | grep -v Warn
