export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2tt
FS_TRACE=1 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || { tail -5 gpurun_out/${T}_bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r2tt_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'stages', d['stages_ms_per_step_rank0'], 'cli', d.get('cli_end_to_end'), flush=True)
PY
grep "reserved ahead\|route+write\|before final\|packFiles total" gpurun_out/${T}_bench.err | tail -8 | cut -c1-200
for i in 1 2; do
  ( TIMEFORMAT="process wall %R s"; time FS_TRACE=1 ./fastore_amd/fastore_pack e -i/tmp/fastore_bench/se10000k.b8 -o/tmp/fastore_bench/cli_t$i -r -f256 -c10 -d8 -w1024 -W1024 ) 2> gpurun_out/${T}_cli_$i.err
  grep -v "^\[trace\] slice [4-9]\|slice 1[0-4]\|^\[bin\]" gpurun_out/${T}_cli_$i.err | cut -c1-230 | tail -16
done
cmp /tmp/fastore_bench/cli_t2.cdata /tmp/fastore_bench/cli.cdata && echo "CLI archives identical"
