cd $GRAFT_REPO_ROOT
L=gpurun_out/exp31.log; : > $L
FS_WATCHDOG=30 timeout 150 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
run() { name=$1; shift
  FS_WATCHDOG=15 timeout 40 env "$@" python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline 2> gpurun_out/exp31.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', 'MB/s', d['value'], 'ms/step', d['ms_per_step'])" >> $L
}
run equal A=1
run w1223 FS_SLICE_WEIGHTS=1,2,2,3
run w1133 FS_SLICE_WEIGHTS=1,1,3,3
run equal A=1
run w1223 FS_SLICE_WEIGHTS=1,2,2,3
run w1133 FS_SLICE_WEIGHTS=1,1,3,3
cat $L
