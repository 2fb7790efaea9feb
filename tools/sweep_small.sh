export RICK_HIP_LIB=rick_amd/lib/librick_hip_abl.so
for cfg in "default" "RICK_U9_MINCHUNKS=1 RICK_U9_SPLIT=1" "RICK_U9_MINCHUNKS=2 RICK_U9_SPLIT=2" "RICK_SPLITK_FIXED=96" "RICK_SPLITK_FIXED=200" "RICK_U9_MINCHUNKS=1 RICK_U9_SPLIT=1 RICK_SPLITK_FIXED=96" "RICK_U9_MINCHUNKS=2 RICK_U9_SPLIT=2 RICK_SPLITK_FIXED=96"; do
  echo "=== $cfg"
  if [ "$cfg" = "default" ]; then python tools/bench_small.py 2>&1 | grep "^B=4.*conv s1\|^B=8.*conv s1" | cut -c1-150; else env $cfg python tools/bench_small.py 2>&1 | grep "^B=4.*conv s1\|^B=8.*conv s1" | cut -c1-150; fi
done
