#!/bin/bash
# kernel trace of a few bench iterations, aggregated by (kernel, grid size): which launch SHAPES the time goes to
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/trace_grid
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out -o tr -- python3 bench.py --steps 8 --warmup 0 --no-fisher --no-cpu-baseline --no-roofline --no-step-times --no-extras > $out/bench.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $out/by_grid.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
    name = r['Kernel_Name'].split('(')[0][:70]
    key = (name, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Workgroup_Size_X']))
    agg[key][0] += 1
    agg[key][1] += d
    tot += d
print(f'total {tot/1e3:.1f} ms, {len(rows)} launches')
for (name, grid, wg), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:150]:
    print(f'{name:70s} grid {grid:6d} x{wg:4d}  n={n:5d}  avg {t/n:8.1f} us  total {t/1e3:8.2f} ms  {100*t/tot:5.2f} %')
PY
find $out -name "*.csv" -size +1M -delete
head -5 $out/by_grid.txt
