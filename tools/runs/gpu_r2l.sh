export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2l
( timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "gather or quality or reproduces or deterministic or sharded or shard_api or cli or seam or larger" ) > gpurun_out/${T}_tests.log 2>&1
tail -5 gpurun_out/${T}_tests.log
run() { # name, env...
  name=$1; shift
  ( env "$@" FS_TRACE=1 FS_WATCHDOG=120 timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cli --no-cpu-baseline ) > gpurun_out/${T}_bench_$name.json 2> gpurun_out/${T}_bench_$name.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/${T}_bench_$name.json').read()); print('$name', d['value'], 'MB/s', d['ms_per_step'], 'ms', d['stages_ms_per_step_rank0'])"
  grep "slice\|batch:" gpurun_out/${T}_bench_$name.err | tail -15 | cut -c1-170
}
run devq
run hostq FS_DEVICE_QUALITY=0
run devq_t16 FS_HOST_THREADS=16
run devq_t32 FS_HOST_THREADS=32
