export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2vv
echo "THP: $(cat /sys/kernel/mm/transparent_hugepage/enabled) defrag: $(cat /sys/kernel/mm/transparent_hugepage/defrag)"
FS_TRACE=1 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || { tail -5 gpurun_out/${T}_bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r2vv_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'stages', d['stages_ms_per_step_rank0'], 'cli', d.get('cli_end_to_end'), flush=True)
PY
for i in 1 2 3 4; do
  A=$(date +%s%3N)
  FS_TRACE=1 ./fastore_amd/fastore_pack e -i/tmp/fastore_bench/se10000k.b8 -o/tmp/fastore_bench/cli_$i -r -f256 -c10 -d8 -w1024 -W1024 2> gpurun_out/${T}_cli_$i.err
  B=$(date +%s%3N)
  echo "run $i: process wall $((B-A)) ms"
  grep "main:\|hipMalloc of\|batch:\|slice [12]/14" gpurun_out/${T}_cli_$i.err | cut -c1-250
done
cmp /tmp/fastore_bench/cli_1.cdata /tmp/fastore_bench/cli_4.cdata && echo archives identical
grep -i "AnonHugePages\|HugePages_Total" /proc/meminfo
