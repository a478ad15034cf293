export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03
# the whole GPU suite on the round's last build, and the reference's thread sweep on the bench library
( timeout -k 10 1000 python -m pytest tests/ -m gpu -x -q ) > gpurun_out/${T}_gpu_suite_last.log 2>&1
tail -4 gpurun_out/${T}_gpu_suite_last.log
( timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --no-cli --no-pe --cpu-sweep ) > gpurun_out/${T}_bench_cpu_sweep.json 2> gpurun_out/${T}_bench_cpu_sweep.err
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_cpu_sweep.json')); print(d['cpu_baseline'])"
