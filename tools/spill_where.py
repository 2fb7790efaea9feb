"""Where a kernel's scratch (spill) traffic sits: per basic-block-loop counts of scratch loads / stores and MFMAs.
usage: python tools/spill_where.py file.s 'mangled-name-substring'"""
import re
import sys

s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):', s, re.M):
    name = m.group(1)
    if sys.argv[2] not in name:
        continue
    body = s[m.end():s.index('.Lfunc_end', m.end())].splitlines()
    loop = 'entry'
    stats = {}
    for l in body:
        mm = re.match(r'(\.LBB\d+_\d+):', l)
        if mm:
            loop = mm.group(1) + (' LOOP' if 'Loop Header' in l else (' in ' + re.search(r'Header=(\w+)', l).group(1) if 'in Loop' in l else ''))
        d = stats.setdefault(loop, [0, 0, 0])
        if 'scratch_load' in l:
            d[0] += 1
        if 'scratch_store' in l:
            d[1] += 1
        if 'v_mfma' in l:
            d[2] += 1
    print(name)
    agg = {}
    for k, v in stats.items():
        key = k.split(' in ')[-1].replace(' LOOP', '') if ('in ' in k or 'LOOP' in k) else 'straight'
        a = agg.setdefault(key, [0, 0, 0])
        for i in range(3):
            a[i] += v[i]
    for k, v in agg.items():
        if any(v):
            print(f'  {k:12s} scratch_load {v[0]:3d}  scratch_store {v[1]:3d}  mfma {v[2]:4d}')
