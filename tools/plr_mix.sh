#!/bin/bash
# kernel mix of the path-length step (graph replay) by kernel name: launches and time per step
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/plr_mix
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o tr -- python3 tools/prof_steps_graph.py ${1:-plr} > $out/run.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $out/mix.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
n = sum(int(r['Calls']) for r in rows)
print(f'total {tot/1e6:.1f} ms, {n} launches (11 step executions + 1 D step: divide by ~11)')
for r in rows[:70]:
    print(f"{r['Name'].split('(')[0][:100]:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['AverageNs'])/1e3:7.1f} us {float(r['Percentage']):5.1f} %")
PY
find $out -name "*.csv" -size +1M -delete
cat $out/mix.txt
