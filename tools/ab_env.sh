#!/bin/bash
# A/B of an environment switch on one box, interleaved: bash tools/ab_env.sh RICK_WGRAD_STREAM 0 1 [n]
var=$1; a=$2; b=$3; n=${4:-2}
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), {k: round(v,2) for k,v in d["step_ms"].items()})'
for i in $(seq 1 $n); do
  env $var=$a timeout 600 python bench.py --no-cpu-baseline --no-extras --no-fisher --no-roofline 2>/dev/null | tail -1 | python -c "$fmt" "$var=$a"
  env $var=$b timeout 600 python bench.py --no-cpu-baseline --no-extras --no-fisher --no-roofline 2>/dev/null | tail -1 | python -c "$fmt" "$var=$b"
done
