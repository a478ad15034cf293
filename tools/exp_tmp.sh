cd $GRAFT_REPO_ROOT
which perf strace ltrace gdb 2>&1 | head -5 > gpurun_out/exp16.log
timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 >> gpurun_out/exp16.log
COPIES=1,3072 timeout 100 python3 tools/ppmd_microbench.py 100000 >> gpurun_out/exp16.log 2>&1
run() { # label, env...
  echo "== $1" >> gpurun_out/exp16.log; shift
  env "$@" FS_TRACE=1 timeout 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/exp16.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('MB/s', d['value'], 'ms/step', d['ms_per_step'], d['stages_ms_per_step'])" >> gpurun_out/exp16.log
  grep -E "batch:" gpurun_out/exp16.err | tail -2 | cut -c1-220 >> gpurun_out/exp16.log
}
run "24 threads" A=1
run "48 threads" FS_HOST_THREADS=48
run "12 threads" FS_HOST_THREADS=12
cat gpurun_out/exp16.log
