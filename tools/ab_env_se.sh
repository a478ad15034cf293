#!/bin/bash
# Warm steps of the BASELINE single-end library under several environments.   tools/ab_env_pe.sh <tag> "<env assignments>" ...
set -u
tag=$1; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
W=${FASTORE_BENCH_DIR:-/tmp/fastore_bench}
out=gpurun_out/${tag}_ab_env_se.txt; : > $out
for e in "$@"; do
env $e python3 - >> $out 2>&1 <<PY
import sys, os, time, subprocess
sys.path.insert(0, os.getcwd())
import bench, fastore_amd
os.makedirs("$W", exist_ok=True)
if not os.path.exists(bench.GEN):
    subprocess.check_call(["g++", "-O2", "-o", bench.GEN, "tools/gen_fastq.cpp"])
cores = len(os.sched_getaffinity(0))
lib = bench.prepare_library("$W", "se10000k", 10000000, 150, 10000000 * 150 // 50, 8, min(cores, 32))[0]
with fastore_amd.Packer(device_id=0) as p:
    ts = []
    for i in range(8):
        t = time.time(); st = p.pack_file(lib, "$W/ab_o"); ts.append(time.time() - t)
    print("$e: steps %s s, front end %.0f ms (sum), searched reads %d" % (" ".join("%.2f" % x for x in ts), st["frontend_ms"], st["matcher_reads"]), flush=True)
PY
done
grep -v "^\[bench" $out
