cd $GRAFT_REPO_ROOT
timeout 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
run() { echo "== $1" >> gpurun_out/exp24.log; shift
  FS_TRACE=1 timeout 200 "$@" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/exp24.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('MB/s', d['value'], 'ms/step', d['ms_per_step'], d['stages_ms_per_step'])" >> gpurun_out/exp24.log
  grep -E "slice ./|batch:" gpurun_out/exp24.err | tail -9 | cut -c1-170 >> gpurun_out/exp24.log
}
run "default" env A=1
run "waves 4096" env FS_MAX_WAVES=4096
run "solo >= 340k" env FS_SOLO_MIN=340000
run "solo >= 340k, waves 4096" env FS_SOLO_MIN=340000 FS_MAX_WAVES=4096
run "solo >= 420k, waves 4096" env FS_SOLO_MIN=420000 FS_MAX_WAVES=4096
run "solo >= 420k, waves 5120" env FS_SOLO_MIN=420000 FS_MAX_WAVES=5120
cat gpurun_out/exp24.log
