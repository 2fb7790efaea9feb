"""HBM traffic of the conv kernels over bench iterations from two rocprofv3 counter passes -> profiles/rNN_pmc_traffic.json
(read by bench.py for roofline.traffic).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o t -- python3 bench.py <ARGS>
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o t -- python3 bench.py <ARGS>
    ARGS = --no-graphs --no-fisher --no-cpu-baseline --no-roofline --no-step-times --steps 16 --warmup 0
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r03_pmc_traffic.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters report KiB; on gfx950
FETCH_SIZE tallies the 128-byte requests of wide (16 B / lane) coalesced reads at 64 bytes, so read bytes = 2 x FETCH_SIZE;
WRITE_SIZE is exact for 16-byte-per-lane stores.  Separate passes: the two counters do not fit the TCC slots together."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys


def kernel_source_hash():
    """sha256 over rick_amd/csrc/* (sorted): bench.py recomputes it and says whether the committed counters still describe
    the kernels it is running."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'rick_amd', 'csrc')
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith(('.hip', '.h')):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), 'rb').read())
    return h.hexdigest()[:16]


def collect(d, counter):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            k = r['Kernel_Name'].split('(')[0].split('<')[0].replace('void ', '')
            tot[k][0] += 1
            tot[k][1] += float(r['Counter_Value'])
    return tot


fetch, write = collect(sys.argv[1], 'FETCH_SIZE'), collect(sys.argv[2], 'WRITE_SIZE')
out = {'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over 16 eager bench iterations, batch 4, 256 px',
       'kernel_source_sha16': kernel_source_hash(),
       'units': 'bytes per launch; reads = 2 x FETCH_SIZE x 1024 (gfx950 correction), writes = WRITE_SIZE x 1024', 'kernels': {}}
fam = {'conv_igemm': ['conv_igemm_kernel', 'conv_igemm_multi_kernel', 'convt2_kernel'], 'conv_wgrad': ['conv_wgrad_kernel', 'conv_wgrad8_kernel']}
for k in sorted(set(fetch) | set(write)):
    n = max(fetch[k][0], write[k][0], 1)
    rd = 2.0 * 1024 * fetch[k][1] / max(fetch[k][0], 1)
    wr = 1024.0 * write[k][1] / max(write[k][0], 1)
    out['kernels'][k] = {'launches': n, 'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr}
for name, members in fam.items():
    n = sum(fetch[m][0] for m in members)
    rd = 2.0 * 1024 * sum(fetch[m][1] for m in members) / max(n, 1)
    wr = 1024.0 * sum(write[m][1] for m in members) / max(sum(write[m][0] for m in members), 1)
    out[name] = {'launches': n, 'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr, 'hbm_bytes_per_launch': rd + wr}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps({k: out[k] for k in fam}, indent=1))
