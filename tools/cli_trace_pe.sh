#!/bin/bash
# The fastore_pack e PROCESS on a paired-end library: wall times of plain runs, the FS_TRACE timeline of one.   tools/cli_trace_pe.sh <tag> [pairs]
set -u
tag=$1; pairs=${2:-25000000}
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
print(bench.prepare_library("$W", "pe%dk" % ($pairs // 1000), $pairs, 150, 2 * $pairs * 150 // 50, 8, min(cores, 32), paired=True))
PY
lib=$W/pe$((pairs/1000))k.b8
out=gpurun_out/${tag}_cli_pe.txt
: > $out
python3 - >> $out <<PY
import subprocess, resource, time
for i in range(2):
    r0 = resource.getrusage(resource.RUSAGE_CHILDREN); t = time.time()
    rc = subprocess.call(["fastore_amd/fastore_pack", "e", "-i$lib", "-o$W/cli_pe_o", "-r", "-f256", "-c10", "-d8", "-w1024", "-W1024", "-z"], stderr=subprocess.DEVNULL)
    dt = time.time() - t; r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    print("run %d: exit %d, %.2f s wall, user %.2f s, system %.2f s, minor faults %d, max RSS %.0f MB" % (i, rc, dt, r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime, r1.ru_minflt - r0.ru_minflt, r1.ru_maxrss / 1024.0), flush=True)
PY
echo "---- FS_TRACE=1 ----" >> $out
s=$(date +%s.%N); FS_TRACE=1 fastore_amd/fastore_pack e -i$lib -o$W/cli_pe_o -r -f256 -c10 -d8 -w1024 -W1024 -z 2>&1 | grep -v "lane teardown\|matcher lane" >> $out; e=$(date +%s.%N)
echo "traced run: $(python3 -c "print('%.2f' % ($e - $s))") s" >> $out
cat $out
