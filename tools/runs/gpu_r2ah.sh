export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2ah
timeout -k 10 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "long_streams" > gpurun_out/${T}_pytest.txt 2>&1; rc=$?
tail -4 gpurun_out/${T}_pytest.txt
[ $rc = 0 ] || exit $rc
python3 bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err || { tail -5 gpurun_out/${T}_bench_default.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r2ah_bench_default.json').read().strip().splitlines()[-1])
print('default bench: value', d['value'], 'ms_per_step', d['ms_per_step'], 'steps', d['steps'], 'cli', d.get('cli_end_to_end'), 'parity', d.get('parity', {}).get('every_block_bit_identical_to_reference'), 'cpu', d.get('cpu_baseline', {}).get('value'), flush=True)
PY
