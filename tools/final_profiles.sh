#!/bin/bash
# Round-end evidence run (one gpurun call): kernel-trace stats of the bench without the Fisher sweep, HBM counters over
# bench iterations, SQ / LDS counters of the three MFMA kernels, the conv and HBM micro-benchmarks.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/final3; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o bench -- python3 bench.py --steps 32 --warmup 0 --no-fisher --no-cpu-baseline --no-roofline --no-step-times --no-extras > $o/stats.log 2>&1
rm -f $o/stats/*/bench_kernel_trace.csv $o/stats/bench_kernel_trace.csv
A="--no-graphs --no-fisher --no-cpu-baseline --no-roofline --no-step-times --no-extras --steps 16 --warmup 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -o t -- python3 bench.py $A > $o/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -o t -- python3 bench.py $A > $o/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $o/pmc_fetch $o/pmc_write $o/r03_pmc_traffic.json > $o/pmc_traffic.txt 2>&1
rm -rf $o/pmc_fetch $o/pmc_write      # (the raw per-dispatch counter tables are > 100 MB: gpurun_out is capped at 64 MiB)
for spec in "conv 512 512 64" "conv 128 128 256" "wgrad 512 512 64 8" "wgrad 512 512 64 4" "wgrad 128 128 256 8" "wgrad_s2 256 512 64 8" "conv_s2 256 512 64 8" "convT2 512 256 64"; do
  tag=$(echo $spec | tr ' ' '_')
  bash tools/pmc_run.sh $o/k_$tag $spec > $o/k_$tag.txt 2>&1
  rm -rf $o/k_$tag/p*/*/*kernel_trace*
done
python3 tools/pmc_conv_json.py $o $o/r03_pmc_conv.json > $o/pmc_conv.txt 2>&1
for d in $o/k_*/; do rm -rf $d; done
python3 tools/bench_conv.py > $o/conv_microbench.txt 2>&1
B=8 python3 tools/bench_conv.py wgrad >> $o/conv_microbench.txt 2>&1
python3 tools/bench_elem.py > $o/hbm_microbench.txt 2>&1
python3 tools/ct2_rounds.py > $o/ct2_rounds.txt 2>&1
[ -x tools/micro/mfma_power ] && tools/micro/mfma_power > $o/mfma_power.txt 2>&1
du -sh $o
