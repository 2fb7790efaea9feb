"""Compact per-kernel resource table (VGPRs / AGPRs / scratch / occupancy / LDS) of one .hip file, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.  usage: python tools/kres.py rick_amd/csrc/conv.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
out = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-Iinclude', '-std=c++17', '-c', src, '-o', '/dev/null',
                      '-Rpass-analysis=kernel-resource-usage'] + sys.argv[3:], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r'\(.*', '', cur).replace('void ', '')
        rows[cur] = {}
        continue
    m = re.search(r'remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill): (\d+)', line)
    if m and cur:
        rows[cur][m.group(1).replace('VGPRs Spill', 'Spill').split()[0]] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print(f"{k:70s} v{v.get('VGPRs', 0):4d} a{v.get('AGPRs', 0):4d} scratch{v.get('ScratchSize', 0):5d} occ{v.get('Occupancy', 0):2d} spill{v.get('Spill', 0):4d}")
