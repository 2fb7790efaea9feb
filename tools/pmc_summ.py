"""Average the per-dispatch counters of rocprofv3 --pmc CSV output per kernel: pmc_summ.py <dir> [name filter]."""
import collections
import csv
import glob
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r['Kernel_Name'].split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name'].split('(')[0][:60]].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for k, cs in rows.items():
    if flt not in k:
        continue
    print(k, f'launches={len(next(iter(cs.values())))}', f'dur_us={sum(dur[k]) / max(len(dur[k]), 1) / 1e3:.1f}' if dur[k] else '')
    for c, v in sorted(cs.items()):
        print(f'    {c:32s} {sum(v) / len(v):16.1f}')
