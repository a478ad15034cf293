export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3y
# QVZ coder: next symbol's word and descriptor requested ahead, double-precision quotients: parity, the --lossy leg
( timeout -k 10 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "qvz or reproduces_reference" ) > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
tail -2 gpurun_out/${T}_tests.log
( timeout -k 10 600 python3 bench.py --quality lossy --steps 3 --warmup 1 --no-cli --no-pe ) > gpurun_out/${T}_bench_se_lossy.json 2> gpurun_out/${T}_bench_se_lossy.err || { tail -5 gpurun_out/${T}_bench_se_lossy.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_se_lossy.json')); print('lossy SE 10 M:', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'], d['cpu_baseline']['value'], d['parity'])"
