#!/bin/bash
# GPU idle time between the kernels of graph-replayed steps: tools/prof_gaps.sh <d|g|r1|plr> [reps]
w=${1:-d}; reps=${2:-12}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_gaps_$w
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gaps_$w -o run -- python3 $root/tools/prof_one_step.py $w $reps > $root/gpurun_out/prof_gaps_$w.log 2>&1
cd $root
tail -1 gpurun_out/prof_gaps_$w.log
f=$(find /tmp/prof_gaps_$w -name '*kernel_trace.csv' | head -1)
python3 - "$f" $reps <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
reps = int(sys.argv[2])
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# the timed region = the last `reps` replays of the profiled step (plus the d steps that precede them are NOT in it): take the tail
# of the trace whose span equals reps x (median step); simpler: split the trace at gaps > 200 us (host syncs) and use the last segment
segs, cur = [], [ev[0]]
for a, b in zip(ev, ev[1:]):
    if b[0] - max(e[1] for e in cur[-64:]) > 200000:
        segs.append(cur); cur = []
    cur.append(b)
segs.append(cur)
seg = max(segs[-3:], key=len)
t0, t1 = seg[0][0], max(e[1] for e in seg)
busy, end = 0, t0
gaps = []
for s, e, _ in seg:
    if s > end:
        gaps.append(s - end); busy += e - s
    else:
        busy += max(0, e - max(s, end))
    end = max(end, e)
span = t1 - t0
print(f'segment: {len(seg)} kernels, span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms ({100*busy/span:.1f} %), idle {100*(1-busy/span):.1f} %')
gaps.sort()
if gaps:
    import statistics
    print(f'gaps: n {len(gaps)}, median {statistics.median(gaps)/1e3:.2f} us, mean {sum(gaps)/len(gaps)/1e3:.2f} us, p90 {gaps[int(.9*len(gaps))]/1e3:.2f} us, sum {sum(gaps)/1e6:.3f} ms')
print(f'per step: {len(seg)/reps:.0f} kernels, {span/reps/1e6:.3f} ms')
PY
