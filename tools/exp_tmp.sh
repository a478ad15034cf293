cd $GRAFT_REPO_ROOT
timeout 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
run() { # label, env...
  echo "== $1" >> gpurun_out/exp15.log; shift
  env "$@" FS_TRACE=1 timeout 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/exp15.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('MB/s', d['value'], 'ms/step', d['ms_per_step'], d['stages_ms_per_step'])" >> gpurun_out/exp15.log
  grep -E "slice ./|batch:" gpurun_out/exp15.err | tail -9 | cut -c1-180 >> gpurun_out/exp15.log
}
run "default 24 threads" A=1
run "threads 32" FS_HOST_THREADS=32
run "threads 48" FS_HOST_THREADS=48
run "threads 64" FS_HOST_THREADS=64
run "threads 96" FS_HOST_THREADS=96
cat gpurun_out/exp15.log
