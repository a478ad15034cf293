#!/bin/bash
# Per-phase clocks of a lone PPMd stream on the GPU box: default build, then the -DFS_WIN_PROFILE and -DFS_SER_PROFILE variants
#   tools/ppmd_phase_profile.sh <tag> [symbols] [libs...]      (libs: paths relative to the repo root; "default" = in-tree build)
set -u
tag=$1; n=${2:-3000000}; shift; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = default ]; then unset FS_LIB FASTORE_AMD_LIB; else export FS_LIB=$PWD/$lib FASTORE_AMD_LIB=$PWD/$lib; fi
  echo "== $name" >> gpurun_out/${tag}_phase.log
  COPIES=${COPIES:-1,1} timeout 300 python3 tools/ppmd_microbench.py $n >> gpurun_out/${tag}_phase.log 2>&1 || exit 1
done
cat gpurun_out/${tag}_phase.log
