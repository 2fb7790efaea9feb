#!/bin/bash
# round 6, item 1: the eight-wave igemm form — parity, same-process A/B per launch, in-situ A/B of bench.py, clock / MFMA-busy counters
out=gpurun_out/w8b; mkdir -p $out
python -m pytest tests/test_gpu_ops.py -q -m gpu -k "eight_wave or conv2d_vs_fp64" -x > $out/tests.txt 2>&1; tail -3 $out/tests.txt
timeout 900 python tools/bench_w8.py check time > $out/w8_b8.txt 2>&1
SCALES=1 B=4 timeout 600 python tools/bench_w8.py time > $out/w8_b4_scales.txt 2>&1
tools/ab_env2.sh RICK_TUNE=0=0 2 > $out/ab_bench.txt 2>&1
export TMPDIR=/tmp
for m in 0 2; do
  export RICK_TUNE=0=$m
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_split_w$m -o a -- python3 tools/pmc_kernel.py conv_split 512 512 64 8 > $out/pmc_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_f32_w$m -o a -- python3 tools/pmc_kernel.py conv 512 512 64 8 >> $out/pmc_$m.log 2>&1
done
unset RICK_TUNE
cat $out/w8_b8.txt $out/ab_bench.txt
