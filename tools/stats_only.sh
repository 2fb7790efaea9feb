cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/final4s; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o bench -- python3 bench.py --steps 32 --warmup 0 --no-fisher --no-cpu-baseline --no-roofline --no-step-times --no-extras > $o/stats.log 2>&1
find $o/stats -name '*kernel_trace.csv' -delete
python bench.py > gpurun_out/bench_final3.json 2> gpurun_out/bench_final3.err
python -c "
import json
d=json.loads(open('gpurun_out/bench_final3.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['nonreg_iteration']['images_per_s'], d['step_ms'], d['roofline']['traffic_source']['counters_describe_current_kernels'])"
