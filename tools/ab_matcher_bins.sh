#!/bin/bash
# The cap of the device window search (FS_MATCHER_BINS: the heaviest n bins of a batch are searched on the device) on ONE paired-end library,
# warm steps.   tools/ab_matcher_bins.sh <tag> [pairs] [caps...]   -> gpurun_out/<tag>_matcher_bins.txt
set -u
tag=$1; pairs=${2:-6000000}; shift; shift
caps=${*:-144 400 1000 4000}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
W=${FASTORE_BENCH_DIR:-/tmp/fastore_bench}
out=gpurun_out/${tag}_matcher_bins.txt; : > $out
for cap in $caps; do
FS_MATCHER_BINS=$cap python3 - >> $out 2>&1 <<PY
import sys, os, time, subprocess
sys.path.insert(0, os.getcwd())
import bench, fastore_amd
os.makedirs("$W", exist_ok=True)
if not os.path.exists(bench.GEN):
    subprocess.check_call(["g++", "-O2", "-o", bench.GEN, "tools/gen_fastq.cpp"])
cores = len(os.sched_getaffinity(0))
lib = bench.prepare_library("$W", "pe%dk" % ($pairs // 1000), $pairs, 150, 2 * $pairs * 150 // 50, 8, min(cores, 32), paired=True)[0]
with fastore_amd.Packer(device_id=0) as p:
    ts = []
    for i in range(4):
        t = time.time(); st = p.pack_file(lib, "$W/pe_o"); ts.append(time.time() - t)
    print("FS_MATCHER_BINS=$cap: steps %s s, front end %.0f ms, searched reads %d, matcher kernels %.0f ms, calls (summed over threads) %.0f ms" % (" ".join("%.2f" % x for x in ts), st["frontend_ms"], st["matcher_reads"], st["matcher_kernel_ms"], st["matcher_call_ms"]), flush=True)
PY
done
grep -v "^\[bench" $out
