#!/bin/bash
# A/B of two builds of the library on one box, interleaved: bash tools/ab_lib.sh rick_amd/lib/librick_hip_old.so [n]
old=$1; n=${2:-2}
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), {k: round(v,2) for k,v in d["step_ms"].items()}, "conv", round(d["roofline"]["conv_family_ms_per_step"],2) if "roofline" in d and d["roofline"] else "")'
for i in $(seq 1 $n); do
  RICK_HIP_LIB=$old python bench.py --no-cpu-baseline --no-extras --no-fisher 2>/dev/null | tail -1 | python -c "$fmt" base
  python bench.py --no-cpu-baseline --no-extras --no-fisher 2>/dev/null | tail -1 | python -c "$fmt" new
done
