#!/bin/bash
# Warm steps of the BASELINE single-end library and of ONE paired-end library with two builds of the library, alternating.
#   tools/ab_lib_steps.sh <tag> <pairs> <lib A (path relative to the repo, or "default")> <lib B> ...
set -u
tag=$1; pairs=$2; shift; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
W=${FASTORE_BENCH_DIR:-/tmp/fastore_bench}
out=gpurun_out/${tag}_ab_lib.txt; : > $out
for round in 1 2; do
for lib in "$@"; do
LIBSEL=$lib python3 - >> $out 2>&1 <<PY
import sys, os, time, subprocess
sys.path.insert(0, os.getcwd())
import bench, fastore_amd
sel = os.environ["LIBSEL"]
lib = None if sel == "default" else fastore_amd.load_library(os.path.join(os.getcwd(), sel))
os.makedirs("$W", exist_ok=True)
if not os.path.exists(bench.GEN):
    subprocess.check_call(["g++", "-O2", "-o", bench.GEN, "tools/gen_fastq.cpp"])
cores = len(os.sched_getaffinity(0))
se = bench.prepare_library("$W", "se10000k", 10000000, 150, 10000000 * 150 // 50, 8, min(cores, 32))[0]
pe = bench.prepare_library("$W", "pe%dk" % ($pairs // 1000), $pairs, 150, 2 * $pairs * 150 // 50, 8, min(cores, 32), paired=True)[0]
for name, path, n in (("SE 10 M", se, 6), ("PE $pairs pairs", pe, 4)):
    kw = dict(device_id=0)
    if lib is not None: kw["lib"] = lib
    with fastore_amd.Packer(**kw) as p:
        ts = []
        for i in range(n):
            t = time.time(); st = p.pack_file(path, "$W/ab_o"); ts.append(time.time() - t)
        print("%s %s: steps %s s, front end %.0f ms (sum)" % (sel, name, " ".join("%.3f" % x for x in ts), st["frontend_ms"]), flush=True)
PY
done
done
grep -v "^\[bench" $out
