#!/bin/bash
# kernel-trace stats of one step type: tools/prof_one_step.sh plr
w=${1:-plr}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_step_$w -o run -- python3 $root/tools/prof_one_step.py $w 16 > $root/gpurun_out/prof_step_$w.log 2>&1
cd $root
tail -1 gpurun_out/prof_step_$w.log
find gpurun_out/prof_step_$w -name '*kernel_trace.csv' -delete
f=$(find gpurun_out/prof_step_$w -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot / 1e6, 'launches', sum(int(r['Calls']) for r in rows))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:40]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {100*float(r['TotalDurationNs'])/tot:5.1f}% {int(r['Calls']):6d} x {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:130]}")
PY
