#!/bin/bash
# The lighter bins' window searches on the device while fewer than FS_SEARCH_SURPLUS threads wait for it (0 = the cap alone: the heaviest
# 6 x threads bins), warm steps of the BASELINE library and of a paired-end one.   tools/ab_search_surplus.sh <tag> [pairs] [values...]
set -u
tag=$1; pairs=${2:-6000000}; shift; shift
vals=${*:-0 8 0 8}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
W=${FASTORE_BENCH_DIR:-/tmp/fastore_bench}
out=gpurun_out/${tag}_search_surplus.txt; : > $out
for v in $vals; do
FS_SEARCH_SURPLUS=$v python3 - >> $out 2>&1 <<PY
import sys, os, time, subprocess
sys.path.insert(0, os.getcwd())
import bench, fastore_amd
os.makedirs("$W", exist_ok=True)
if not os.path.exists(bench.GEN):
    subprocess.check_call(["g++", "-O2", "-o", bench.GEN, "tools/gen_fastq.cpp"])
cores = len(os.sched_getaffinity(0))
se = bench.prepare_library("$W", "se10000k", 10000000, 150, 10000000 * 150 // 50, 8, min(cores, 32))[0]
pe = bench.prepare_library("$W", "pe%dk" % ($pairs // 1000), $pairs, 150, 2 * $pairs * 150 // 50, 8, min(cores, 32), paired=True)[0]
for name, lib, n in (("SE 10 M", se, 6), ("PE $pairs pairs", pe, 4)):
    with fastore_amd.Packer(device_id=0) as p:
        ts = []; r0 = 0
        for i in range(n):
            t = time.time(); st = p.pack_file(lib, "$W/ab_o"); ts.append(time.time() - t)
            if i == n - 2: r0 = st["matcher_reads"]
        print("FS_SEARCH_SURPLUS=$v %s: steps %s s, searched reads a step %d, front end %.0f ms (all steps)" % (name, " ".join("%.3f" % x for x in ts), st["matcher_reads"] - r0, st["frontend_ms"]), flush=True)
PY
done
grep -v "^\[bench" $out
