#!/bin/bash
# same-box A/B: generator layers on-the-fly (default) vs split-image hand-over (RICK_GSPLIT=1)
n=${1:-2}
for i in $(seq 1 $n); do
  python bench.py --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('G on-the-fly', round(d['value'],2), {k: round(v,2) for k,v in d['step_ms'].items()})"
  RICK_GSPLIT=1 python bench.py --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('G images    ', round(d['value'],2), {k: round(v,2) for k,v in d['step_ms'].items()})"
done
