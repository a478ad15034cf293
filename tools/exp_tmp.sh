cd $GRAFT_REPO_ROOT
timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/exp12.log
run() { # label, env...
  echo "== $1" >> gpurun_out/exp12.log; shift
  env "$@" FS_TRACE=1 timeout 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/exp12.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('MB/s', d['value'], 'ms/step', d['ms_per_step'], d['stages_ms_per_step'])" >> gpurun_out/exp12.log
  grep -E "slice ./|batch:|flush|packFiles total|route" gpurun_out/exp12.err | tail -9 | cut -c1-200 >> gpurun_out/exp12.log
}
run "default (solo on)" A=1
run "solo off" FS_SOLO_MIN=0
run "solo on, waves 4096" FS_MAX_WAVES=4096
run "solo on, waves 5120" FS_MAX_WAVES=5120
run "solo min 150000, waves 4096" FS_SOLO_MIN=150000 FS_MAX_WAVES=4096
cat gpurun_out/exp12.log
