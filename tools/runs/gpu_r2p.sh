export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2p
( timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "matcher or gather or quality_streams or reproduces" ) > gpurun_out/${T}_tests.log 2>&1
tail -5 gpurun_out/${T}_tests.log




run() { # name, env...
  name=$1; shift
  ( env "$@" FS_TRACE=1 FS_WATCHDOG=120 timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cli --no-cpu-baseline ) > gpurun_out/${T}_bench_$name.json 2> gpurun_out/${T}_bench_$name.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/${T}_bench_$name.json').read()); print('$name', d['value'], 'MB/s', d['ms_per_step'], 'ms', d['stages_ms_per_step_rank0'])"
  grep "slice\|batch:\|matcher" gpurun_out/${T}_bench_$name.err | tail -15 | cut -c1-170 | grep -v "slice [4-9]/\|slice 1[0-2]/"
}
run devm
run hostm FS_DEVICE_MATCHER=0
