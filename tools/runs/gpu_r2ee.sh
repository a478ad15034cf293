export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2ee
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/${T}_tests.log 2>&1
tail -4 gpurun_out/${T}_tests.log
( FS_TRACE=1 FS_WATCHDOG=120 timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cli --no-cpu-baseline ) > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python3 -c "
import json,sys
d=json.loads(open('gpurun_out/${T}_bench.json').read()); print(d['value'], 'MB/s', d['ms_per_step'], 'ms', d['stages_ms_per_step_rank0'], d['h2d_bytes_per_step'])"
grep "packFiles total" gpurun_out/${T}_bench.err | tr '\n' ' '; echo
grep "slice 1/\|slice 2/\|batch:" gpurun_out/${T}_bench.err | tail -3 | cut -c1-170
