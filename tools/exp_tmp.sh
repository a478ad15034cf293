cd $GRAFT_REPO_ROOT
timeout 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
run() { echo "== $1" >> gpurun_out/exp21.log; shift
  FS_TRACE=1 timeout 200 "$@" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/exp21.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('MB/s', d['value'], 'ms/step', d['ms_per_step'], d['stages_ms_per_step'])" >> gpurun_out/exp21.log
  grep -E "slice ./|batch:|close|total" gpurun_out/exp21.err | tail -11 | cut -c1-200 >> gpurun_out/exp21.log
}
run "default" env A=1
run "default again" env A=1
run "waves 3584" env FS_MAX_WAVES=3584
cat gpurun_out/exp21.log
