#!/bin/bash
# Resource usage of the fastore_pack e PROCESS on the BASELINE library (faults, system time): tools/cli_rusage.sh <tag>
set -u
tag=$1
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
W=${FASTORE_BENCH_DIR:-/tmp/fastore_bench}
python3 - <<PY
import sys, os
sys.path.insert(0, os.getcwd())
import bench, subprocess
os.makedirs("$W", exist_ok=True)
if not os.path.exists(bench.GEN):
    subprocess.check_call(["g++", "-O2", "-o", bench.GEN, "tools/gen_fastq.cpp"])
cores = len(os.sched_getaffinity(0))
bench.prepare_library("$W", "se10000k", 10000000, 150, 10000000 * 150 // 50, 8, min(cores, 32))
PY
lib=$W/se10000k.b8
out=gpurun_out/${tag}_rusage.txt
{ echo "THP: $(cat /sys/kernel/mm/transparent_hugepage/enabled) defrag: $(cat /sys/kernel/mm/transparent_hugepage/defrag)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; } > $out
python3 - >> $out <<PY
import subprocess, resource, time
for i in range(3):
    r0 = resource.getrusage(resource.RUSAGE_CHILDREN); t = time.time()
    subprocess.call(["fastore_amd/fastore_pack", "e", "-i$lib", "-o$W/cli_o", "-r", "-f256", "-c10", "-d8", "-w1024", "-W1024"], stderr=subprocess.DEVNULL)
    dt = time.time() - t; r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    print("run %d: %.2f s wall, user %.2f s, system %.2f s, minor faults %d, major %d, voluntary switches %d, involuntary %d, max RSS %.0f MB" % (i, dt, r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime, r1.ru_minflt - r0.ru_minflt, r1.ru_majflt - r0.ru_majflt, r1.ru_nvcsw - r0.ru_nvcsw, r1.ru_nivcsw - r0.ru_nivcsw, r1.ru_maxrss / 1024.0))
PY
cat $out
