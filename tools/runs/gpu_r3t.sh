export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3t
# windowed range coder (small alphabets): parity, then what it does to the PPMd forms (registers) and to the --reduced step
( timeout -k 10 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "rc_device or reproduces_reference or every_kernel_form or qvz_device" ) > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
tail -2 gpurun_out/${T}_tests.log
FS_WAVES=1 COPIES=3072 timeout -k 10 200 python3 tools/ppmd_microbench.py 3000000 2>&1 | tee gpurun_out/${T}_micro_w1.txt | head -1
FS_WAVES=2 COPIES=1 timeout -k 10 120 python3 tools/ppmd_microbench.py 7000000 2>&1 | tee gpurun_out/${T}_micro_w2.txt | head -1
( timeout -k 10 400 python3 bench.py --steps 4 --warmup 2 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench_lossless.json 2> gpurun_out/${T}_bench_lossless.err || { tail -5 gpurun_out/${T}_bench_lossless.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_lossless.json')); print('lossless SE:', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'])"
( FS_TRACE=1 timeout -k 10 600 python3 bench.py --quality reduced --steps 3 --warmup 1 --no-cli --no-pe ) > gpurun_out/${T}_bench_se_reduced.json 2> gpurun_out/${T}_bench_se_reduced.err || { tail -5 gpurun_out/${T}_bench_se_reduced.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_se_reduced.json')); print('reduced SE 10 M:', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'], d['cpu_baseline']['value'], d['parity'])"
grep "slice [1-3]/" gpurun_out/${T}_bench_se_reduced.err | tail -3 | cut -c1-230
