"""Summarise the per-kernel counter passes of tools/final_profiles.sh (k_<mode>_<ci>_<co>_<r>[_<B>]/p1..p4) into one JSON:
MFMA-busy, instruction mix per MFMA, LDS conflict share, waits, HBM bytes.   usage: pmc_conv_json.py <final dir> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

root, outp = sys.argv[1], sys.argv[2]
out = {'source': 'tools/final_profiles.sh: four rocprofv3 --pmc passes per kernel (tools/pmc_run.sh: SQ timing; LDS / instruction mix; '
                 'FETCH_SIZE; WRITE_SIZE) on tools/pmc_kernel.py <mode ci co r [B]>, 6 launches each, 3x3 kernels, batch 4 unless the tag '
                 'ends in _8; SQ_INSTS_VALU includes the MFMAs (subtracted here); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / '
                 '(GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); HBM bytes per the guide (reads = 2 x FETCH_SIZE KiB on gfx950, writes = WRITE_SIZE KiB)',
       'kernels': {}}
for d in sorted(glob.glob(os.path.join(root, 'k_*'))):
    if not os.path.isdir(d):
        continue
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            vals[r['Kernel_Name'].split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
    best = max((k for k in vals if 'conv' in k and 'reduce' not in k and 'pack' not in k and 'split_pack' not in k), key=lambda k: sum(vals[k].get('SQ_INSTS_MFMA', [0])), default=None)
    if best is None:
        continue
    c = {k: sum(v) / len(v) for k, v in vals[best].items()}
    mf = max(c.get('SQ_INSTS_MFMA', 0.0), 1.0)
    ent = {'kernel': best, 'mfma_busy_frac': c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(c.get('GRBM_GUI_ACTIVE', 1) / 8 * 1024, 1),
           'mfma_busy_frac_of_sq_busy': c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(32 * c.get('SQ_BUSY_CYCLES', 1), 1),
           'lds_bank_conflict_frac_of_lds_cycles': c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1),
           'non_mfma_valu_insts_per_mfma': (c.get('SQ_INSTS_VALU', 0) - c.get('SQ_INSTS_MFMA', 0)) / mf,
           'salu_insts_per_mfma': c.get('SQ_INSTS_SALU', 0) / mf, 'lds_insts_per_mfma': c.get('SQ_INSTS_LDS', 0) / mf,
           'vmem_rd_insts_per_mfma': c.get('SQ_INSTS_VMEM_RD', 0) / mf,
           'wait_any_frac_of_wave_cycles': c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1),
           'wait_inst_any_frac_of_wave_cycles': c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1),
           'hbm_read_MB': 2 * 1024 * c.get('FETCH_SIZE', 0) / 1e6, 'hbm_write_MB': 1024 * c.get('WRITE_SIZE', 0) / 1e6, 'raw': c}
    out['kernels'][os.path.basename(d)[2:]] = ent
json.dump(out, open(outp, 'w'), indent=1)
for k, e in out['kernels'].items():
    print(f"{k:26s} {e['kernel'][:46]:46s} busy {e['mfma_busy_frac']:.3f} ({e['mfma_busy_frac_of_sq_busy']:.3f})  conflicts {e['lds_bank_conflict_frac_of_lds_cycles']:.3f}  "
          f"VALU/SALU/LDS per MFMA {e['non_mfma_valu_insts_per_mfma']:.2f}/{e['salu_insts_per_mfma']:.2f}/{e['lds_insts_per_mfma']:.2f}  "
          f"HBM r/w MB {e['hbm_read_MB']:.0f}/{e['hbm_write_MB']:.0f}")
