cd $GRAFT_REPO_ROOT
timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/exp23.log
run() { echo "== $1" >> gpurun_out/exp23.log; shift
  FS_TRACE=1 timeout 200 "$@" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/exp23.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('MB/s', d['value'], 'ms/step', d['ms_per_step'], d['stages_ms_per_step'])" >> gpurun_out/exp23.log
  grep -E "slice 1/|slice 8/|batch:|close|total|route" gpurun_out/exp23.err | tail -7 | cut -c1-200 >> gpurun_out/exp23.log
}
run "default" env A=1
run "default again" env A=1
cat gpurun_out/exp23.log
