"""ISA gate and survey for the packed-fp32 question (round 4's modbank_fwd fault: the compiler's `v_pk_fma_f32` form of a
reduce-over-lanes kernel returned wrong sums next to a busy neighbour process; cause not established, scalar FMAs never failed).

  python tools/check_isa.py --gate     fails (exit 1) if a kernel on the no-packing list contains a packed fp32 instruction.
                                       Run by __graft_entry__.build(): nothing stops hipcc's SLP vectoriser from pairing scalar
                                       FMAs again after an innocent edit, so the build checks the ISA it produced.
  python tools/check_isa.py --survey   every kernel of the library that combines packed fp32 arithmetic with a cross-lane
                                       operation (ds_bpermute / ds_swizzle / DPP / permlane / readlane), with instruction counts —
                                       the list tests/test_gpu_determinism.py's busy-neighbour test is built from
                                       (profiles/r06_isa_packed_crosslane.txt).

Device assembly comes from `hipcc -S --cuda-device-only` with the Makefile's flags (same code generation as the shipped
objects)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'rick_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
CXXFILT = next((p for p in ('/opt/rocm/lib/llvm/bin/llvm-cxxfilt', '/usr/bin/c++filt') if os.path.exists(p)), 'c++filt')
FLAGS = ['-O3', '-fPIC', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'), '-std=c++17', '-S', '--cuda-device-only']
PACKED = re.compile(r'^\s*(v_pk_fma_f32|v_pk_mul_f32|v_pk_add_f32)\b')
CROSS = re.compile(r'^\s*(ds_bpermute_b32|ds_permute_b32|ds_swizzle_b32|v_permlane\w*|v_readlane_b32|v_writelane_b32|v_\w+_dpp)\b')
# kernels whose cross-lane sums were wrong in the packed form, or share its pattern (LDS-broadcast operand x FMA chain -> wave sum)
NO_PACKING = {'modulation.hip': ['modbank_fwd_kernel'], 'linear.hip': ['linear_fwd_kernel', 'linear_dgrad_kernel', 'linear_wgrad_kernel']}


PER_FILE = {'linear.hip': ['-fno-slp-vectorize']}      # (rick_amd/csrc/Makefile: target-specific CXXFLAGS)


def device_asm(src):
    r = subprocess.run([HIPCC] + FLAGS + PER_FILE.get(src, []) + [os.path.join(CSRC, src), '-o', '-'], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(f'hipcc -S {src} failed:\n{r.stderr[-2000:]}')
    return r.stdout


def kernels(asm):
    """{mangled name: [instruction lines]} for every .amdhsa kernel of one translation unit"""
    names = set(re.findall(r'^\s*\.amdhsa_kernel\s+(\S+)', asm, re.M))
    out = {}
    for m in re.finditer(r'^(\w+):\s*(?:;.*)?$', asm, re.M):
        if m.group(1) in names:
            end = asm.find('.Lfunc_end', m.end())
            out[m.group(1)] = asm[m.end():end].splitlines()
    return out


def demangle(n):
    try:
        return subprocess.run([CXXFILT, n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


def gate():
    bad = []
    for src, subs in NO_PACKING.items():
        ks = kernels(device_asm(src))
        for sub in subs:
            hit = [k for k in ks if sub in k]
            if not hit:
                bad.append(f'{src}: kernel {sub} not found (renamed? update tools/check_isa.py)')
            for k in hit:
                n = sum(1 for l in ks[k] if PACKED.match(l))
                if n:
                    bad.append(f'{src}: {demangle(k)[:80]} contains {n} packed fp32 instructions (v_pk_*_f32); its products must stay '
                               f'scalar (MB_FMAC inline asm in modulation.hip, -fno-slp-vectorize for linear.hip): see modbank_fwd_kernel')
    if bad:
        raise SystemExit('ISA gate FAILED:\n  ' + '\n  '.join(bad))
    print('ISA gate ok: no packed fp32 arithmetic in', ', '.join(s for v in NO_PACKING.values() for s in v))


def survey(path=None):
    """aggregated per (source, kernel template): variants, packed / cross-lane instruction ranges, kinds"""
    agg = {}
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith('.hip')):
        for k, body in kernels(device_asm(src)).items():
            npk = sum(1 for l in body if PACKED.match(l))
            cross = [CROSS.match(l).group(1) for l in body if CROSS.match(l)]
            if npk and cross:
                base = re.sub(r'^void ', '', demangle(k)).split('(')[0].split('<')[0]
                e = agg.setdefault((src, base), [0, [], [], set()])
                e[0] += 1
                e[1].append(npk)
                e[2].append(len(cross))
                e[3] |= set('dpp' if c.endswith('_dpp') else c for c in cross)
    lines = ['# tools/check_isa.py --survey: kernels of librick_hip.so that combine packed fp32 arithmetic (v_pk_fma/mul/add_f32) with a',
             '# cross-lane operation, per kernel template.  source | kernel | variants | packed fp32 instr. (min-max) | cross-lane instr. (min-max) | kinds',
             '# The combination is the rule, not the exception: the fp16 hi/lo split and the epilogues are packed multiplies, the block',
             '# maxima / sums go through ds_bpermute, v_readlane / v_writelane are SGPR spills.  What failed in round 4 was narrower — packed',
             '# FMA ACCUMULATORS feeding a wave sum (modbank_fwd_kernel) — and that pattern is what the build gate keeps scalar:',
             '# ' + ', '.join(s for v in NO_PACKING.values() for s in v) + ' (none of them may appear below).',
             '# Every family below is re-evaluated next to a busy neighbour process by tests/test_gpu_determinism.py /',
             '# tools/stress_ops.py (0 differences in 2 000 launches each, round 5; repeated in round 6: profiles/r06_stress_ops.txt).']
    for (src, base), (n, pk, cr, kinds) in sorted(agg.items()):
        lines.append(f'{src:16s} | {base:28s} | {n:3d} | {min(pk)}-{max(pk)} | {min(cr)}-{max(cr)} | {",".join(sorted(kinds))}')
    text = '\n'.join(lines) + '\n'
    if path:
        open(path, 'w').write(text)
    print(text, end='')
    return agg


if __name__ == '__main__':
    if '--survey' in sys.argv:
        i = sys.argv.index('--survey')
        survey(sys.argv[i + 1] if len(sys.argv) > i + 1 else None)
    else:
        gate()
