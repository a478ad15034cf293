export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2o
( timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "matcher or gather or quality_streams or reproduces" ) > gpurun_out/${T}_tests.log 2>&1
tail -5 gpurun_out/${T}_tests.log
FS_TWO_WAVE=1 FS_LIB=build/libfastore_amd_prof.so COPIES=1 timeout 600 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/${T}_prof_3M_two.txt 2>&1
cat gpurun_out/${T}_prof_3M_two.txt
FS_TWO_WAVE=1 COPIES=1 timeout 600 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/${T}_micro_3M_two.txt 2>&1
cat gpurun_out/${T}_micro_3M_two.txt
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
