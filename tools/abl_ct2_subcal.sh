# time of a convt2 block with 4 / 2 of its tile's 8 fragment columns against a whole-tile block
export RICK_HIP_LIB=rick_amd/lib/librick_hip_abl.so
for shape in "512,512,32,1 11,11,1,1" "512,256,64,1 11,11,1,1" "256,128,128,1 10,12,1,1" "512,256,64,2 11,11,1,1"; do
  set -- $shape
  for q in 1 2 4; do
    echo "== $1 tile $2 subq $q"
    CT2_ONLY=$1 RICK_CT2_TILE=$2 RICK_CT2_SUBQ=$q timeout 120 python tools/ct2_rounds.py 2>&1 | grep rounds
  done
done
