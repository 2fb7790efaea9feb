#!/bin/bash
# same-box A/B of the eight-wave weight-gradient form in the experiment build (RICK_WGRAD8=0: four-wave form), alternating runs
export RICK_HIP_LIB=rick_amd/lib/librick_hip_abl.so
mkdir -p gpurun_out/ab
for i in 1 2; do
  RICK_WGRAD8=1 python bench.py --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 > gpurun_out/ab/wg8_on_$i.json
  RICK_WGRAD8=0 python bench.py --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 > gpurun_out/ab/wg8_off_$i.json
done
python - <<'PY'
import json
for i in (1, 2):
    for tag in ('on', 'off'):
        d = json.load(open(f'gpurun_out/ab/wg8_{tag}_{i}.json'))
        print(f'wgrad8 {tag:3s}', round(d['value'], 2), 'img/s', round(d['ms_per_step'], 3), 'ms', {k: round(v, 2) for k, v in d.get('step_ms', {}).items()})
PY
