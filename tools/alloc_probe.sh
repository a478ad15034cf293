#!/bin/bash
# Device allocations of processes that start right behind each other: one piece against many.   tools/alloc_probe.sh <tag>
set -u
tag=$1
cd "$(dirname "$0")/.."
mkdir -p gpurun_out build
hipcc --offload-arch=gfx950 -O2 -o build/hip_alloc_probe tools/hip_alloc_probe.hip || exit 1
out=gpurun_out/${tag}_alloc_probe.txt; : > $out
for cfg in "17.8 1" "17.8 8" "17.8 64" "34.4 1" "34.4 8" "34.4 64"; do
  echo "== $cfg, 16 processes back to back" >> $out
  for i in $(seq 1 16); do build/hip_alloc_probe $cfg 200 >> $out 2>&1; done
done
cat $out
