export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2k
./build/salu_probe > gpurun_out/${T}_salu_probe.txt 2>&1; cat gpurun_out/${T}_salu_probe.txt
run() { # name, env...
  name=$1; shift
  ( env "$@" FS_TRACE=1 FS_WATCHDOG=120 timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cli --no-cpu-baseline ) > gpurun_out/${T}_bench_$name.json 2> gpurun_out/${T}_bench_$name.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/${T}_bench_$name.json').read()); print('$name', d['value'], 'MB/s', d['ms_per_step'], 'ms', d['stages_ms_per_step_rank0'])"
  grep "slice" gpurun_out/${T}_bench_$name.err | tail -14 | cut -c1-160
}
run old8 FS_PIPELINE_SLICES=8
run new14_q16 GPU_MAX_HW_QUEUES=16 FS_PIPELINE_LANES=14
run new14_q24 GPU_MAX_HW_QUEUES=24 FS_PIPELINE_LANES=14
