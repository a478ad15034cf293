export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2t
run() { # name, env...
  name=$1; shift
  ( env "$@" FS_TRACE=1 FS_WATCHDOG=120 timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cli --no-cpu-baseline ) > gpurun_out/${T}_bench_$name.json 2> gpurun_out/${T}_bench_$name.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/${T}_bench_$name.json').read()); print('$name', d['value'], 'MB/s', d['ms_per_step'], 'ms', d['stages_ms_per_step_rank0'])"
  grep "packFiles total" gpurun_out/${T}_bench_$name.err | tr '\n' ' '; echo
}
run m144
run m96 FS_MATCHER_BINS=96
run m192 FS_MATCHER_BINS=192
run m0 FS_MATCHER_BINS=0
