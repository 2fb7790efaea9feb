#!/bin/bash
# Round-6 evidence run (one gpurun call): the bench line, kernel-trace stats of the bench command, HBM counters over bench iterations
# (separate --pmc passes, no other trace domains), SQ / LDS counters of the MFMA kernels, micro-benchmarks on device time.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/final6; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o bench -- python3 bench.py --steps 32 --warmup 0 --no-fisher --no-cpu-baseline --no-roofline --no-step-times --no-extras > $o/stats.log 2>&1
find $o/stats -name '*kernel_trace.csv' -delete
A="--no-graphs --no-fisher --no-cpu-baseline --no-roofline --no-step-times --no-extras --steps 16 --warmup 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -o t -- python3 bench.py $A > $o/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -o t -- python3 bench.py $A > $o/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $o/pmc_fetch $o/pmc_write $o/r06_pmc_traffic.json > $o/pmc_traffic.txt 2>&1
rm -rf $o/pmc_fetch $o/pmc_write
cp $o/r06_pmc_traffic.json profiles/r06_pmc_traffic.json      # (bench.py reads the newest committed counter file: the line below carries these)
python3 bench.py > $o/bench_line.json 2> $o/bench_stderr.log
for spec in "conv 512 512 64 8" "conv_split 512 512 64 8" "conv 128 128 256 8" "conv_split 128 128 256 8" "wgrad 512 512 64 8" "wgrad_split 512 512 64 8" "wgrad_s2 256 512 64 8" "wgrad_s2_split 256 512 64 8" "conv_s2 256 512 64 8" "conv_s2_split 256 512 64 8" "convT2 512 256 64 8" "convT2_split 512 256 64 8"; do
  tag=$(echo $spec | tr ' ' '_')
  bash tools/pmc_run.sh $o/k_$tag $spec > $o/k_$tag.txt 2>&1
  find $o/k_$tag -name '*kernel_trace*' -delete
done
python3 tools/pmc_conv_json.py $o $o/r06_pmc_conv.json > $o/pmc_conv.txt 2>&1
for d in $o/k_*/; do rm -rf $d; done
python3 tools/bench_elem.py 2>&1 | grep -v "Warning\|_warn_once" > $o/hbm_microbench.txt
python3 tools/bench_thin.py 2>&1 | grep -v "Warning\|_warn_once" >> $o/hbm_microbench.txt
python3 tools/bench_small.py 2>&1 | grep "^B=" > $o/small_conv_microbench.txt
RICK_TUNE=3=0 python3 tools/bench_elem.py 2>&1 | grep -v "Warning\|_warn_once" > $o/hbm_microbench_tile8.txt      # (the 8 x 8 x 64 FIR tile, for the A/B)
bash tools/prof_one_step.sh plr > $o/step_mix_plr.txt 2>&1
bash tools/prof_one_step.sh r1 > $o/step_mix_r1.txt 2>&1
python3 tools/stability.py > $o/stability.txt 2>&1
du -sh $o
tail -1 $o/bench_line.json | head -c 600
