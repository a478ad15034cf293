export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2aa
( timeout 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "ppmd or reproduces or deterministic" ) > gpurun_out/${T}_tests.log 2>&1
tail -3 gpurun_out/${T}_tests.log
FS_TWO_WAVE=1 FS_LIB=build/libfastore_amd_prof.so COPIES=1 timeout 600 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/${T}_prof_3M_two.txt 2>&1
cat gpurun_out/${T}_prof_3M_two.txt
FS_TWO_WAVE=1 COPIES=1 timeout 600 python3 tools/ppmd_microbench.py 7000000 > gpurun_out/${T}_micro_7M_two.txt 2>&1
cat gpurun_out/${T}_micro_7M_two.txt

( FS_TRACE=1 FS_WATCHDOG=120 timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cli --no-cpu-baseline ) > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python3 -c "
import json,sys
d=json.loads(open('gpurun_out/${T}_bench.json').read()); print(d['value'], 'MB/s', d['ms_per_step'], 'ms', d['stages_ms_per_step_rank0'])"
grep "packFiles total" gpurun_out/${T}_bench.err | tr '\n' ' '; echo
grep "slice 1/\|slice 2/\|batch:" gpurun_out/${T}_bench.err | tail -3 | cut -c1-170
