#!/bin/bash
# The fastore_pack e process on the BASELINE library with the box's preloaded library (orderly teardown) and without it (the process leaves without
# teardown): wall times.   tools/cli_exit_paths.sh <tag>
set -u
tag=$1
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
W=${FASTORE_BENCH_DIR:-/tmp/fastore_bench}
python3 - <<PY
import sys, os, subprocess
sys.path.insert(0, os.getcwd())
import bench
os.makedirs("$W", exist_ok=True)
if not os.path.exists(bench.GEN):
    subprocess.check_call(["g++", "-O2", "-o", bench.GEN, "tools/gen_fastq.cpp"])
cores = len(os.sched_getaffinity(0))
bench.prepare_library("$W", "se10000k", 10000000, 150, 10000000 * 150 // 50, 8, min(cores, 32))
PY
lib=$W/se10000k.b8
out=gpurun_out/${tag}_exit_paths.txt
echo "LD_PRELOAD=${LD_PRELOAD:-}" > $out
for round in 1 2 3; do
  for mode in preload nopreload; do
    s=$(date +%s.%N)
    if [ $mode = preload ]; then fastore_amd/fastore_pack e -i$lib -o$W/cli_o -r -f256 -c10 -d8 -w1024 -W1024 2>/dev/null; else env -u LD_PRELOAD fastore_amd/fastore_pack e -i$lib -o$W/cli_o -r -f256 -c10 -d8 -w1024 -W1024 2>/dev/null; fi
    rc=$?; e=$(date +%s.%N)
    echo "$mode run $round: exit $rc, $(python3 -c "print('%.2f' % ($e - $s))") s" >> $out
  done
done
cat $out
