export TMPDIR=/tmp
mkdir -p gpurun_out
FS_TWO_WAVE=1 timeout 600 python3 tools/ppmd_contention.py 3000000 600000 > gpurun_out/r2cc_contention.txt 2>&1
MAXW=1536 FS_TWO_WAVE=1 COPIES=1024,3000 timeout 600 python3 tools/ppmd_contention.py 3000000 600000 >> gpurun_out/r2cc_contention.txt 2>&1
cat gpurun_out/r2cc_contention.txt
( FS_TRACE=1 FS_WATCHDOG=120 timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cli --no-cpu-baseline ) > gpurun_out/r2cc_bench.json 2> gpurun_out/r2cc_bench.err
python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r2cc_bench.json').read()); print(d['value'], 'MB/s', d['ms_per_step'], 'ms', d['stages_ms_per_step_rank0'])"
grep "packFiles total" gpurun_out/r2cc_bench.err | tr '\n' ' '; echo
grep "route+write" gpurun_out/r2cc_bench.err | tr '\n' ' '; echo
