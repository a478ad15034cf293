export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3m
# free-list heads in LDS: parity, lone stream, serial-path clocks, bench
( timeout -k 10 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "every_kernel_form or ppmd_device or reproduces_reference" ) > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
tail -2 gpurun_out/${T}_tests.log
FS_WAVES=2 COPIES=1 timeout -k 10 120 python3 tools/ppmd_microbench.py 7000000 > gpurun_out/${T}_micro_7M_w2.txt 2>&1; cat gpurun_out/${T}_micro_7M_w2.txt
FS_WAVES=1 COPIES=3072 timeout -k 10 200 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/${T}_micro_3M_w1_3072.txt 2>&1; cat gpurun_out/${T}_micro_3M_w1_3072.txt
FS_WAVES=2 FS_LIB=build/libfastore_amd_prof.so COPIES=1 timeout -k 10 120 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/${T}_prof_3M_w2.txt 2>&1; cat gpurun_out/${T}_prof_3M_w2.txt
( timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench.json')); print('SE', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'])"
