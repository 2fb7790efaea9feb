#!/bin/bash
# same-box A/B of one environment switch, alternating runs:  tools/ab_env2.sh RICK_NO_ADJOINT_DOT=1 [pairs]
sw=$1; n=${2:-2}
mkdir -p gpurun_out/ab
for i in $(seq 1 $n); do
  python bench.py --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 > gpurun_out/ab/default_$i.json
  env $sw python bench.py --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 > gpurun_out/ab/switch_$i.json
done
python - "$sw" $n <<'PY'
import json, sys
sw, n = sys.argv[1], int(sys.argv[2])
for i in range(1, n + 1):
    for tag in ('default', 'switch'):
        d = json.load(open(f'gpurun_out/ab/{tag}_{i}.json'))
        print(f'{tag if tag == "default" else sw:28s}', round(d['value'], 2), 'img/s', round(d['ms_per_step'], 3), 'ms', {k: round(v, 2) for k, v in d.get('step_ms', {}).items()})
PY
