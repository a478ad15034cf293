export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2bb
( FS_TRACE=1 FS_WATCHDOG=120 timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cli --no-cpu-baseline ) > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python3 -c "
import json,sys
d=json.loads(open('gpurun_out/${T}_bench.json').read()); print(d['value'], 'MB/s', d['ms_per_step'], 'ms', d['stages_ms_per_step_rank0'])"
grep "packFiles total" gpurun_out/${T}_bench.err | tr '\n' ' '; echo
grep "route+write" gpurun_out/${T}_bench.err | tr '\n' ' '; echo
( time FS_TRACE=1 ./fastore_amd/fastore_pack e -i/tmp/fastore_bench/se10000k.b8 -o/tmp/fastore_bench/cli_t -r -f256 -c10 -d8 -w1024 -W1024 ) > gpurun_out/${T}_cli.out 2> gpurun_out/${T}_cli.err
grep -v "slice\|\[bin\]" gpurun_out/${T}_cli.err | tail -22 | cut -c1-200
