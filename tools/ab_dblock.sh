#!/bin/bash
# same-box A/B: per-layer discriminator blocks (RICK_NO_DBLOCK=1) vs the one-node split-image blocks
n=${1:-2}
for i in $(seq 1 $n); do
  RICK_NO_DBLOCK=1 python bench.py --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('per-layer', round(d['value'],2), {k: round(v,2) for k,v in d['step_ms'].items()})"
  python bench.py --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-node ', round(d['value'],2), {k: round(v,2) for k,v in d['step_ms'].items()})"
done
