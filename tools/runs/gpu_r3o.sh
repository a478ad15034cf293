export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3o
# device-side unpack of the bases (fs_unpack_planes): parity, then the step with it and without
( timeout -k 10 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "unpack or matcher or reproduces_reference" ) > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
tail -2 gpurun_out/${T}_tests.log
for M in 1 0 1 0; do
  ( FS_DEVICE_UNPACK=$M timeout -k 10 400 python3 bench.py --steps 4 --warmup 2 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench_unpack$M.json 2> gpurun_out/${T}_bench_unpack$M.err || { tail -5 gpurun_out/${T}_bench_unpack$M.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_unpack$M.json')); print('unpack $M: SE', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'], {k: v for k, v in d.get('stats', {}).items() if 'matcher' in k})"
done
