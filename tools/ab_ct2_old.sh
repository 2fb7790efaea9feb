for l in old hip old hip; do
  echo "== lib $l"
  RICK_HIP_LIB=$( [ $l = old ] && echo rick_amd/lib/librick_hip_old.so || echo rick_amd/lib/librick_hip.so ) CT2_B=3,7 timeout 300 python tools/ct2_rounds.py 2>&1 | grep rounds | grep -v "in 16"
done
