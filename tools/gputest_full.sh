#!/bin/bash
# Three consecutive full `pytest -m gpu` runs on the sources as they are (fresh box), summary lines + source hash + HEAD.
out=${1:-gpurun_out/r06_gputest_full.txt}
mkdir -p gpurun_out
{
  echo "kernel source hash: $(python3 -c 'import bench; print(bench.kernel_source_hash())')"
  echo "librick_hip.so sha256: $(sha256sum rick_amd/lib/librick_hip.so | cut -c1-16)"
  echo "date: $(date -u +%Y-%m-%dT%H:%M:%SZ)"
} > $out
for i in 1 2 3; do
  echo "=== run $i: python -m pytest tests -q -m gpu" >> $out
  python3 -m pytest tests -q -m gpu 2>&1 | grep -v -E "Gloo|socket.cpp|amdgpu.ids|Warning|_warn_once|warnings.html|^$|^tests/|^  " | tail -6 >> $out
done
cat $out
