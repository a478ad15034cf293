export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/r03_smoke.log 2>&1; echo "exit $?"; tail -3 gpurun_out/r03_smoke.log
( timeout -k 10 400 python3 bench.py --steps 2 --warmup 1 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/r03_bench_quick.json 2> gpurun_out/r03_bench_quick.err; python3 -c "
import json; d=json.load(open('gpurun_out/r03_bench_quick.json')); print('lossless SE:', d['value'], d['ms_per_step'])"
