export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2s
run() { # name, env...
  name=$1; shift
  ( env "$@" FS_TRACE=1 FS_WATCHDOG=120 timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cli --no-cpu-baseline ) > gpurun_out/${T}_bench_$name.json 2> gpurun_out/${T}_bench_$name.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/${T}_bench_$name.json').read()); print('$name', d['value'], 'MB/s', d['ms_per_step'], 'ms', d['stages_ms_per_step_rank0'], d['other_kernels'])"
  grep "slice\|batch:\|matcher" gpurun_out/${T}_bench_$name.err | tail -16 | cut -c1-170 | grep -v "slice [4-9]/\|slice 1[0-2]/"
}
run m250 FS_MATCHER_BINS=250
run m400 FS_MATCHER_BINS=400
run all_w2048 FS_MATCHER_BINS=100000 FS_MAX_WAVES=2048
run all_w1024 FS_MATCHER_BINS=100000 FS_MAX_WAVES=1024
run m144_w2048 FS_MATCHER_BINS=144 FS_MAX_WAVES=2048
