#!/bin/bash
# A/B of the working tree against a copy of an earlier state in _ab/ (same box, interleaved runs):
#   git worktree add -f _ab <rev>; rm _ab/.git; make -C _ab/rick_amd/csrc;  gpurun -- 'bash tools/ab_bench.sh 3'
n=${1:-2}
mkdir -p gpurun_out
for i in $(seq 1 $n); do
  (cd _ab && python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', round(d['value'],2), {k: round(v,2) for k,v in d['step_ms'].items()})")
  python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', round(d['value'],2), {k: round(v,2) for k,v in d['step_ms'].items()})"
done
