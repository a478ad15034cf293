export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2gg
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python3 -c "
import json,sys
d=json.loads(open('gpurun_out/${T}_bench.json').read()); print(d['value'], 'MB/s', d['ms_per_step'], 'ms', d['cli_end_to_end'])"
for i in 1 2 3; do ( time ./fastore_amd/fastore_pack e -i/tmp/fastore_bench/se10000k.b8 -o/tmp/fastore_bench/cli_t$i -r -f256 -c10 -d8 -w1024 -W1024 ) 2>&1 | grep real; done
( FS_TRACE=1 ./fastore_amd/fastore_pack e -i/tmp/fastore_bench/se10000k.b8 -o/tmp/fastore_bench/cli_t4 -r -f256 -c10 -d8 -w1024 -W1024 ) 2>&1 | grep -v "slice\|\[bin\]" | tail -8 | cut -c1-200
