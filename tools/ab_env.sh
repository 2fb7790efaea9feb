#!/bin/bash
# Same-box A/B of an environment switch (interleaved bench runs): tools/ab_env.sh RICK_NO_DEMOD_BANK 3
var=$1; n=${2:-2}
for i in $(seq 1 $n); do
  env $var=1 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=1', round(d['value'],2), {k: round(v,2) for k,v in d['step_ms'].items()})"
  python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default ', round(d['value'],2), {k: round(v,2) for k,v in d['step_ms'].items()})"
done
