export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2uu
python3 - <<'PY'
import sys, time; sys.path.insert(0, '.')
import bench, os
os.makedirs('/tmp/fastore_bench', exist_ok=True)
b, size = bench.prepare_library('/tmp/fastore_bench', 'se10000k', 10_000_000, 150, 30_000_000, 8, min(os.cpu_count(), 32))
PY
for i in 1 2 3 4 5; do
  if [ $i = 5 ]; then export FS_ORDERLY_EXIT=1; fi
  A=$(date +%s%3N)
  FS_TRACE=1 ./fastore_amd/fastore_pack e -i/tmp/fastore_bench/se10000k.b8 -o/tmp/fastore_bench/cli_$i -r -f256 -c10 -d8 -w1024 -W1024 2> gpurun_out/${T}_cli_$i.err
  B=$(date +%s%3N)
  echo "run $i: shell start $A, shell end $B, process wall $((B-A)) ms"
  grep "main:\|hipMalloc of\|batch:\|set-up\|slice [12]/14\|matcher so far" gpurun_out/${T}_cli_$i.err | cut -c1-250
  echo ----
done
cmp /tmp/fastore_bench/cli_1.cdata /tmp/fastore_bench/cli_5.cdata && echo archives identical
