#!/bin/bash
# A/B of the paired-end front end on the GPU box: ONE library of <pairs> pairs, the warm step under FS_DEVICE_MATES=0 / 2 / 3 (and whatever else is given)
#   tools/ab_mates.sh <tag> <pairs> <mode> [<mode> ...]
set -u
tag=$1; pairs=$2; shift; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for mode in "$@"; do
  echo "== FS_DEVICE_MATES=$mode" >> gpurun_out/${tag}_mates.log
  FS_DEVICE_MATES=$mode timeout 500 python3 bench.py --paired --reads $pairs --steps 3 --warmup 1 --no-cli --no-cpu-baseline --in-process 2>> gpurun_out/${tag}_mates.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('MB/s', d['value'], 'ms/step', d['ms_per_step'], d['each_step_ms'], 'stages', d['stages_ms_per_step_rank0'], 'mates', d['other_kernels']['fs_match_mates'])" >> gpurun_out/${tag}_mates.log
done
cat gpurun_out/${tag}_mates.log
