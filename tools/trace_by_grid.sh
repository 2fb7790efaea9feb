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
# the launch sequence of the last ~1.5 iterations, in order (for fusion hunting)
seq = sorted(rows, key=lambda r: int(r['Start_Timestamp']))[-1800:]
with open(sys.argv[1].rsplit('/', 1)[0] + '/../sequence.txt', 'w') as f:
    for r in seq:
        f.write(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3:8.1f} us  grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d}  {r['Kernel_Name'].split('(')[0][:90]}\n")
# idle time between consecutive kernels over the last half of the trace (graph replay): gap histogram
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
ev = ev[len(ev) // 2:]
busy = sum(e - s for s, e in ev) * 1e-3
span = (max(e for s, e in ev) - ev[0][0]) * 1e-3
gaps = [max(0, ev[i + 1][0] - max(e for s, e in ev[max(0, i - 3):i + 1])) * 1e-3 for i in range(len(ev) - 1)]
import statistics
print(f'last half: {len(ev)} launches, busy {busy/1e3:.2f} ms of span {span/1e3:.2f} ms = {100*busy/span:.1f} %; '
      f'gap median {statistics.median(gaps):.2f} us, mean {sum(gaps)/len(gaps):.2f} us, total {sum(gaps)/1e3:.2f} ms; '
      f'gaps > 20 us: {sum(1 for g in gaps if g > 20)} totalling {sum(g for g in gaps if g > 20)/1e3:.2f} ms')
for (name, grid, wg), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:150]:
    print(f'{name:70s} grid {grid:6d} x{wg:4d}  n={n:5d}  avg {t/n:8.1f} us  total {t/1e3:8.2f} ms  {100*t/tot:5.2f} %')
PY
find $out -name "*.csv" -size +1M -delete
head -5 $out/by_grid.txt
