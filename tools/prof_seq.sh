#!/bin/bash
# kernel sequence of one graph-replayed step around a named kernel: tools/prof_seq.sh <d|g> <substring>
w=${1:-d}; pat=${2:-copyBuffer}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_seq_$w
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seq_$w -o run -- python3 $root/tools/prof_one_step.py $w 4 > $root/gpurun_out/prof_seq_$w.log 2>&1
cd $root
f=$(find /tmp/prof_seq_$w -name '*kernel_trace.csv' | head -1)
python3 - "$f" "$pat" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# last step = kernels after the last masked_adam but one
adam = [i for i, e in enumerate(ev) if 'masked_adam' in e[2]]
seg = ev[adam[-2] + 1:adam[-1] + 1]
print('kernels in the last step:', len(seg), 'span ms', (seg[-1][1] - seg[0][0]) / 1e6)
short = lambda n: n.split('(')[0][-60:]
for i, e in enumerate(seg):
    if pat in e[2]:
        print(i, '|', short(seg[i - 1][2]) if i else '-', '->', short(e[2]), f'{(e[1]-e[0])/1e3:.1f}us', '->', short(seg[i + 1][2]) if i + 1 < len(seg) else '-')
c = collections.Counter(short(e[2]) for e in seg)
dur = collections.defaultdict(float)
for e in seg: dur[short(e[2])] += (e[1] - e[0]) / 1e3
small = [(k, n, dur[k]) for k, n in c.items() if dur[k] / n < 12]
print('small kernels (avg < 12 us):', sum(n for _, n, _ in small), 'launches', round(sum(t for _, _, t in small)), 'us')
for k, n, t in sorted(small, key=lambda x: -x[2])[:28]:
    print(f'{n:4d} x {t/n:5.1f} us = {t:6.0f} us  {k}')
PY
