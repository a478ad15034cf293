cd $GRAFT_REPO_ROOT
timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/exp14.log
COPIES=1,3072 timeout 100 python3 tools/ppmd_microbench.py 100000 >> gpurun_out/exp14.log 2>&1
run() { # label, env...
  echo "== $1" >> gpurun_out/exp14.log; shift
  env "$@" timeout 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/exp14.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('MB/s', d['value'], 'ms/step', d['ms_per_step'], d['stages_ms_per_step'])" >> gpurun_out/exp14.log
}
run "default" A=1
run "8 slices" FS_PIPELINE_SLICES=8
run "default again" A=1
cat gpurun_out/exp14.log
head -2 tools/pmc_passes_default.txt > /tmp/p2.txt
timeout 200 bash tools/pmc_microbench.sh v5 /tmp/p2.txt > /dev/null 2>&1
