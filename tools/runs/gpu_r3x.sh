export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3x
# windowed range coder with the next window's rows fetched behind the write-back: parity, the --reduced legs
( timeout -k 10 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "rc_device or reproduces_reference" ) > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
tail -2 gpurun_out/${T}_tests.log
for i in 1 2; do
( FS_TRACE=1 timeout -k 10 600 python3 bench.py --quality reduced --steps 4 --warmup 1 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench_se_reduced.json 2> gpurun_out/${T}_bench_se_reduced.err || { tail -5 gpurun_out/${T}_bench_se_reduced.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_se_reduced.json')); print('reduced SE 10 M:', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'])"
grep "slice [1-3]/" gpurun_out/${T}_bench_se_reduced.err | tail -3 | cut -c1-230
done
