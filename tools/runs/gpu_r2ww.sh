export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2ww
python3 - <<'PY'
import sys, time; sys.path.insert(0, '.')
import bench, os
os.makedirs('/tmp/fastore_bench', exist_ok=True)
bench.prepare_library('/tmp/fastore_bench', 'se10000k', 10_000_000, 150, 30_000_000, 8, min(os.cpu_count(), 32))
PY
# warm the box's device memory (the first processes on a fresh box wait for the driver to clear what it hands out)
for i in 1 2 3; do ./fastore_amd/fastore_pack e -i/tmp/fastore_bench/se10000k.b8 -o/tmp/fastore_bench/w -r -f256 -c10 -d8 -w1024 -W1024; done
for P in 1 0 1 0 1 0; do
  A=$(date +%s%3N)
  FS_PAGEABLE_STAGING=$P FS_TRACE=1 ./fastore_amd/fastore_pack e -i/tmp/fastore_bench/se10000k.b8 -o/tmp/fastore_bench/cli_$P -r -f256 -c10 -d8 -w1024 -W1024 2> gpurun_out/${T}_cli_$P.err
  B=$(date +%s%3N)
  echo "pageable=$P: process wall $((B-A)) ms; $(grep 'main: context' gpurun_out/${T}_cli_$P.err | cut -c1-120); $(grep 'slice 1/14' gpurun_out/${T}_cli_$P.err | cut -c30-200)"
done
cmp /tmp/fastore_bench/cli_1.cdata /tmp/fastore_bench/cli_0.cdata && echo archives identical
echo "== PE 1.5 M pairs: bench (one-pass mate scan)"
python3 bench.py --paired --reads 1500000 --steps 3 --warmup 1 --no-cli > gpurun_out/${T}_pe.json 2> gpurun_out/${T}_pe.err || { tail -5 gpurun_out/${T}_pe.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r2ww_pe.json').read().strip().splitlines()[-1])
print('PE value', d['value'], 'ms_per_step', d['ms_per_step'], 'stages', d['stages_ms_per_step_rank0'], 'cpu', d.get('cpu_baseline',{}).get('value'), 'parity', d.get('parity'), flush=True)
PY
