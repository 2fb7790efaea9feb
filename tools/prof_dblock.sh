#!/bin/bash
# kernel-trace stats of a short bench run (graphs) -> gpurun_out/prof_$1/ and a top-45 table
tag=${1:-dblock}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o run -- python3 $root/bench.py --no-cpu-baseline --no-roofline --no-extras --no-step-times --steps 16 --warmup 2 > $root/gpurun_out/prof_$tag.log 2>&1
cd $root
tail -3 gpurun_out/prof_$tag.log
find gpurun_out/prof_$tag -name '*kernel_trace.csv' -delete
f=$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot / 1e6)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:45]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {100*float(r['TotalDurationNs'])/tot:5.1f}% {int(r['Calls']):6d} x {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:140]}")
PY
