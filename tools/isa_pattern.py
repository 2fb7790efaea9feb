"""Instruction pattern of a kernel's MFMA-heavy basic blocks (M mfma, r/w LDS read/write, G global load, D LDS-DMA,
v VALU, s SALU, | s_waitcnt, B barrier, J branch, X scratch).  usage: python tools/isa_pattern.py file.hip 'mangled substring' [min mfma]"""
import re
import subprocess
import sys

src, sub = sys.argv[1], sys.argv[2]
minm = int(sys.argv[3]) if len(sys.argv) > 3 else 24
asm = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-Iinclude', '-std=c++17', '-S', '--cuda-device-only', src,
                      '-o', '-'], capture_output=True, text=True).stdout
for m in re.finditer(r'^(_Z\w+):', asm, re.M):
    if sub not in m.group(1):
        continue
    body = asm[m.end():asm.index('.Lfunc_end', m.end())].splitlines()
    print(m.group(1))
    blocks, cur = [], ['entry']
    for l in body:
        if re.match(r'\.LBB\d+_\d+:', l):
            blocks.append(cur)
            cur = [l]
        else:
            cur.append(l)
    blocks.append(cur)
    for bl in blocks:
        n = sum('v_mfma' in l for l in bl)
        if n < minm:
            continue
        out = []
        for l in bl[1:]:
            t = l.strip().split()[0] if l.strip() else ''
            c = ('M' if t.startswith('v_mfma') else 'r' if t.startswith('ds_read') else 'w' if t.startswith('ds_write') else
                 'D' if t.startswith('global_load_lds') else 'G' if t.startswith('global_load') else '|' if t.startswith('s_waitcnt') else
                 'B' if t.startswith('s_barrier') else 'J' if t.startswith(('s_cbranch', 's_branch')) else 'X' if t.startswith('scratch') else
                 'v' if t.startswith('v_') else 's' if t.startswith('s_') else '')
            out.append(c)
        q = ''.join(out)
        print(' ', bl[0].split(':')[0], 'mfma', n, 'instr', len(q), 'scratch', q.count('X'))
        for i in range(0, len(q), 150):
            print('    ', q[i:i + 150])
