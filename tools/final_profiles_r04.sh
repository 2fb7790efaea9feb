#!/bin/bash
# Round-4 evidence run (one gpurun call): kernel-trace stats of the bench, HBM counters over bench iterations, SQ / LDS counters of
# the MFMA kernels on fp32 operands and on split images, micro-benchmarks.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/final4; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o bench -- python3 bench.py --steps 32 --warmup 0 --no-fisher --no-cpu-baseline --no-roofline --no-step-times --no-extras > $o/stats.log 2>&1
find $o/stats -name '*kernel_trace.csv' -delete
A="--no-graphs --no-fisher --no-cpu-baseline --no-roofline --no-step-times --no-extras --steps 16 --warmup 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -o t -- python3 bench.py $A > $o/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -o t -- python3 bench.py $A > $o/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $o/pmc_fetch $o/pmc_write $o/r04_pmc_traffic.json > $o/pmc_traffic.txt 2>&1
rm -rf $o/pmc_fetch $o/pmc_write
for spec in "conv 512 512 64 8" "conv_split 512 512 64 8" "conv 128 128 256 8" "conv_split 128 128 256 8" "wgrad 512 512 64 8" "wgrad_split 512 512 64 8" "wgrad 128 128 256 8" "wgrad_split 128 128 256 8" "wgrad_s2 256 512 64 8" "wgrad_s2_split 256 512 64 8" "conv_s2 256 512 64 8" "conv_s2_split 256 512 64 8" "convT2 512 256 64 8" "convT2_split 512 256 64 8" "conv 512 512 64 4" "wgrad 512 512 64 4"; do
  tag=$(echo $spec | tr ' ' '_')
  bash tools/pmc_run.sh $o/k_$tag $spec > $o/k_$tag.txt 2>&1
  find $o/k_$tag -name '*kernel_trace*' -delete
done
python3 tools/pmc_conv_json.py $o $o/r04_pmc_conv.json > $o/pmc_conv.txt 2>&1
for d in $o/k_*/; do rm -rf $d; done
python3 tools/bench_conv.py > $o/conv_microbench.txt 2>&1
B=8 python3 tools/bench_conv.py wgrad >> $o/conv_microbench.txt 2>&1
B=8 python3 tools/bench_split.py wgrad conv > $o/split_microbench.txt 2>&1
python3 tools/bench_split_g.py >> $o/split_microbench.txt 2>&1
python3 tools/bench_producers.py > $o/producers_microbench.txt 2>&1
python3 tools/bench_elem.py > $o/hbm_microbench.txt 2>&1
python3 tools/bench_thin.py >> $o/hbm_microbench.txt 2>&1
python3 tools/bench_actbwd.py >> $o/hbm_microbench.txt 2>&1
du -sh $o
