export TMPDIR=/tmp
mkdir -p gpurun_out
python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/r2ak_bench.json 2> gpurun_out/r2ak_bench.err || { tail -5 gpurun_out/r2ak_bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r2ak_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'steps', d['steps'])
PY
ls -la /tmp/fastore_bench | grep "out_" | head
python3 bench.py > gpurun_out/r2ak_bench_default.json 2> gpurun_out/r2ak_bench_default.err || { tail -5 gpurun_out/r2ak_bench_default.err; exit 1; }
tail -c 700 gpurun_out/r2ak_bench_default.json
