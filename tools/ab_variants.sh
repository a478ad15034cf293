#!/bin/bash
# A/B of library builds on the GPU box: PPMd micro-benchmark (3072 x 100 k symbols) and the bench step per variant.
#   tools/ab_variants.sh <tag> <lib1> <lib2> ...     (paths relative to the repo root; "default" = in-tree build)
set -u
tag=$1; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1     # prepares (and caches) the libraries
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = default ]; then unset FS_LIB FASTORE_AMD_LIB; else export FS_LIB=$PWD/$lib FASTORE_AMD_LIB=$PWD/$lib; fi
  echo "== $name" >> gpurun_out/${tag}_ab.log
  COPIES=1,3072 timeout 120 python3 tools/ppmd_microbench.py 100000 >> gpurun_out/${tag}_ab.log 2>&1
  for sl in ${SLICES:-5}; do
  FS_PIPELINE_SLICES=$sl timeout 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>> gpurun_out/${tag}_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('slices $sl: bench MB/s', d['value'], 'ms/step', d['ms_per_step'], 'stages', d['stages_ms_per_step'])" >> gpurun_out/${tag}_ab.log
  done
done
cat gpurun_out/${tag}_ab.log
