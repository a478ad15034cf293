#!/bin/bash
# Where a paired-end step's host time goes: the stage clocks of EVERY bin's front end (FS_BIN_TRACE=2) summed over the bins of one warm pack.
#   tools/pe_frontend_stages.sh <tag> [pairs]   -> gpurun_out/<tag>_se_stages.txt
set -u
tag=$1; pairs=${2:-6000000}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
W=${FASTORE_BENCH_DIR:-/tmp/fastore_bench}
FS_BIN_TRACE=2 python3 - 2> gpurun_out/${tag}_se_stages.err > gpurun_out/${tag}_se_stages.txt <<PY
import sys, os, time, subprocess
sys.path.insert(0, os.getcwd())
import bench, fastore_amd
os.makedirs("$W", exist_ok=True)
if not os.path.exists(bench.GEN):
    subprocess.check_call(["g++", "-O2", "-o", bench.GEN, "tools/gen_fastq.cpp"])
cores = len(os.sched_getaffinity(0))
lib = bench.prepare_library("$W", "se%dk" % ($pairs // 1000), $pairs, 150, $pairs * 150 // 50, 8, min(cores, 32))[0]
print("library", lib, flush=True)
with fastore_amd.Packer(device_id=0) as p:
    for i in range(2):
        sys.stderr.write("==== pack %d\n" % i); sys.stderr.flush()
        t = time.time(); st = p.pack_file(lib, "$W/se_o"); print("pack %d: %.2f s  frontend_ms %.0f" % (i, time.time() - t, st.frontend_ms), flush=True)
PY
python3 - >> gpurun_out/${tag}_se_stages.txt <<PY
import re
lines = open("gpurun_out/${tag}_se_stages.err").read().split("==== pack 1")[-1].splitlines()
names = ["nodes + sort", "match table + device search", "top-level tree", "contigs + sub-trees + emission", "mate searches + their streams"]
tot = [0.0] * 5; n = 0; big = []
for l in lines:
    m = re.match(r"\[bin\] (\d+) records: nodes \+ sort ([\d.]+) ms, match table \+ device search ([\d.]+) ms \(pre (\d)\), top-level tree ([\d.]+) ms, contigs \+ sub-trees \+ emission ([\d.]+) ms, mate searches \+ their streams ([\d.]+) ms", l)
    if not m: continue
    v = [float(m.group(i)) for i in (2, 3, 5, 6, 7)]
    for i in range(5): tot[i] += v[i]
    n += 1; big.append((int(m.group(1)), v))
print("bins", n, "summed stage ms over the bins of the second pack:")
for i in range(5): print("  %-36s %10.0f ms  %5.1f %%" % (names[i], tot[i], 100 * tot[i] / max(sum(tot), 1e-9)))
print("  total %.0f ms" % sum(tot))
big.sort(reverse=True)
for r, v in big[:5]: print("  heaviest bin %d records:" % r, v)
PY
cat gpurun_out/${tag}_se_stages.txt
