#!/bin/bash
# same-box A/B of op.wgrad_overlap (RICK_WGRAD_OVERLAP=1 = sunk weight gradients on a second stream), alternating runs
mkdir -p gpurun_out/r05
for i in 1 2; do
  RICK_WGRAD_OVERLAP=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05/ab_overlap_on_$i.json
  python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05/ab_overlap_off_$i.json
done
python - <<'PY'
import json
for f in ('on_1', 'off_1', 'on_2', 'off_2'):
    d = json.load(open(f'gpurun_out/r05/ab_overlap_{f}.json'))
    print(f, round(d['value'], 2), 'img/s', round(d['ms_per_step'], 3), 'ms', {k: round(v, 2) for k, v in d['step_ms'].items()}, 'nonreg', round(d['nonreg_iteration']['median_ms'], 2))
PY
