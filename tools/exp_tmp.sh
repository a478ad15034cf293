cd $GRAFT_REPO_ROOT
timeout 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
run() { # label, env...
  echo "== $1" >> gpurun_out/exp11.log; shift
  env "$@" FS_TRACE=1 timeout 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/exp11.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('MB/s', d['value'], 'ms/step', d['ms_per_step'], d['stages_ms_per_step'])" >> gpurun_out/exp11.log
  grep -E "slice ./|batch:|flush|packFiles total" gpurun_out/exp11.err | tail -8 | cut -c1-200 >> gpurun_out/exp11.log
}
run "default" A=1
run "waves 4096" FS_MAX_WAVES=4096
run "waves 5120" FS_MAX_WAVES=5120
run "threads 32" FS_HOST_THREADS=32
run "threads 48" FS_HOST_THREADS=48
run "threads 64" FS_HOST_THREADS=64
for w in 0 4096 5120; do
  echo "== microbench max_waves $w" >> gpurun_out/exp11.log
  COPIES=1,$((w>0?w:3072)),$((2*(w>0?w:3072))) timeout 100 python3 tools/ppmd_microbench.py 100000 $w >> gpurun_out/exp11.log 2>&1
done
cat gpurun_out/exp11.log
timeout 300 bash tools/pmc_microbench.sh v4 tools/pmc_passes_default.txt > /dev/null 2>&1
