export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3z
# the arena pool grows to 32 MiB slots when a launch first needs them (not at creation): the --reduced leg, the lossless leg, then the whole GPU suite
( FS_TRACE=1 timeout -k 10 600 python3 bench.py --quality reduced --steps 3 --warmup 1 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench_se_reduced.json 2> gpurun_out/${T}_bench_se_reduced.err || { tail -5 gpurun_out/${T}_bench_se_reduced.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_se_reduced.json')); print('reduced SE 10 M:', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'])"
grep "arena pool" gpurun_out/${T}_bench_se_reduced.err | head -3
( FS_TRACE=1 timeout -k 10 400 python3 bench.py --steps 4 --warmup 2 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench_lossless.json 2> gpurun_out/${T}_bench_lossless.err || { tail -5 gpurun_out/${T}_bench_lossless.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_lossless.json')); print('lossless SE:', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'])"
grep "arena pool" gpurun_out/${T}_bench_lossless.err | head -3
( timeout -k 10 1000 python -m pytest tests/ -m gpu -x -q ) > gpurun_out/r03_gpu_suite_last.log 2>&1
tail -4 gpurun_out/r03_gpu_suite_last.log
