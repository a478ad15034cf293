#!/bin/bash
# The fastore_pack e PROCESS on the BASELINE library (10 M x 150 bp SE, prepared with the reference's tools as bench.py does): wall times of
# plain runs and the FS_TRACE timeline of one.   tools/cli_trace.sh <tag> [reads]
set -u
tag=$1; reads=${2:-10000000}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
W=${FASTORE_BENCH_DIR:-/tmp/fastore_bench}
python3 - <<PY
import sys, os
sys.path.insert(0, os.getcwd())
import bench
os.makedirs("$W", exist_ok=True)
if not os.path.exists(bench.GEN):
    import subprocess; subprocess.check_call(["g++", "-O2", "-o", bench.GEN, "tools/gen_fastq.cpp"])
cores = len(os.sched_getaffinity(0))
print(bench.prepare_library("$W", "se%dk" % ($reads // 1000), $reads, 150, $reads * 150 // 50, 8, min(cores, 32)), cores, "cores")
PY
lib=$W/se$((reads/1000))k.b8
out=gpurun_out/${tag}_cli.txt
: > $out
for i in 1 2 3 4 5; do
  s=$(date +%s.%N); fastore_amd/fastore_pack e -i$lib -o$W/cli_o -r -f256 -c10 -d8 -w1024 -W1024 2>/dev/null; rc=$?; e=$(date +%s.%N)
  echo "run $i: exit $rc, $(python3 -c "print('%.2f' % ($e - $s))") s" >> $out
done
echo "---- FS_TRACE=1 ----" >> $out
s=$(date +%s.%N); FS_TRACE=1 fastore_amd/fastore_pack e -i$lib -o$W/cli_o -r -f256 -c10 -d8 -w1024 -W1024 2>> $out; e=$(date +%s.%N)
echo "traced run: $(python3 -c "print('%.2f' % ($e - $s))") s" >> $out
echo "---- FS_TRACE=1 FS_BIN_TRACE=40000 (stage clocks of the heaviest bins) ----" >> $out
FS_TRACE=1 FS_BIN_TRACE=40000 fastore_amd/fastore_pack e -i$lib -o$W/cli_o -r -f256 -c10 -d8 -w1024 -W1024 2>&1 | grep -E "^\[bin\]|slice 1/|main:|device_create: up|record arrays|library 0" | head -150 >> $out
echo "---- a warm context with the host's window scan (FS_DEVICE_MATCHER=0): stage clocks of the heaviest bins ----" >> $out
FS_DEVICE_MATCHER=0 FS_BIN_TRACE=40000 python3 - 2>&1 <<PY3 | grep -E "nodes \+ sort|pack" | head -14 >> $out
import sys, os, time
sys.path.insert(0, os.getcwd())
import fastore_amd
with fastore_amd.Packer(device_id=0) as p:
    for i in range(2):
        t = time.time(); p.pack_file("$lib", "$W/py_o"); print("pack %d: %.2f s" % (i, time.time() - t), flush=True)
PY3
echo "---- the same in a warm context (python: one Packer, three packs; FS_BIN_TRACE on the last) ----" >> $out
FS_BIN_TRACE=40000 python3 - >> $out 2>&1 <<PY2
import sys, os, time
sys.path.insert(0, os.getcwd())
import fastore_amd
with fastore_amd.Packer(device_id=0) as p:
    for i in range(3):
        t = time.time(); p.pack_file("$lib", "$W/py_o"); print("pack %d: %.2f s" % (i, time.time() - t), flush=True)
PY2
cat $out
