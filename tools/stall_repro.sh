#!/bin/bash
# Repeats the default bench step under FS_WATCHDOG to catch the device stall seen with more than four concurrent launches
# (profiles/r01_stall_bisect.txt).  Edit the run lines for other settings (FS_PIPELINE_SLICES,
# GPU_MAX_HW_QUEUES, FASTORE_AMD_LIB=<variant build>).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/exp30.log; : > $L
FS_WATCHDOG=30 timeout 400 python3 bench.py --steps 2 --warmup 1 > gpurun_out/r01g_bench.json 2> gpurun_out/r01g_bench.err; echo "bench rc=$?" >> $L
for i in 1 2 3 4 5 6 7 8; do
  T0=$(date +%s)
  FS_WATCHDOG=15 timeout 60 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/exp30.json 2> gpurun_out/exp30_$i.err
  rc=$?
  echo "default run $i rc=$rc secs=$(( $(date +%s) - T0 )) $(python3 -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/exp30.json').read()); print('MB/s', d['value'])
except Exception as e: print('no json')")" >> $L
  grep -E "watchdog" gpurun_out/exp30_$i.err | head -12 | cut -c1-200 >> $L
done
FS_WATCHDOG=30 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01g_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r01g_bench_under_rocprof.json 2> gpurun_out/r01g_stats.err
python3 tools/pmc_summary.py stats gpurun_out/r01g_stats > gpurun_out/r01g_kernel_stats.json
FS_WATCHDOG=30 timeout 300 python3 -m pytest tests -m gpu -x -q -k "deterministic or reproduces_reference or fresh_library" > gpurun_out/r01g_pytest.log 2>&1; echo "pytest rc=$?" >> $L; tail -2 gpurun_out/r01g_pytest.log >> $L
cat $L; cat gpurun_out/r01g_bench.json
