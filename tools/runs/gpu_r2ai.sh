export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2ai
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
import bench, os
os.makedirs('/tmp/fastore_bench', exist_ok=True)
bench.prepare_library('/tmp/fastore_bench', 'se10000k', 10_000_000, 150, 30_000_000, 8, min(os.cpu_count(), 32))
PY
for i in 1 2 3; do ./fastore_amd/fastore_pack e -i/tmp/fastore_bench/se10000k.b8 -o/tmp/fastore_bench/w -r -f256 -c10 -d8 -w1024 -W1024; sleep 3; done
for S in 0 1 0 1 0 1; do
  sleep 4
  A=$(date +%s%3N)
  if [ $S = 1 ]; then export FS_SYNC_DEVICE=1; else unset FS_SYNC_DEVICE; fi
  FS_TRACE=1 ./fastore_amd/fastore_pack e -i/tmp/fastore_bench/se10000k.b8 -o/tmp/fastore_bench/cli_$S -r -f256 -c10 -d8 -w1024 -W1024 2> gpurun_out/${T}_cli_$S.err
  B=$(date +%s%3N)
  echo "sync_device=$S: process wall $((B-A)) ms; $(grep 'main: context' gpurun_out/${T}_cli_$S.err | cut -c1-110); $(grep 'slice 1/14' gpurun_out/${T}_cli_$S.err | cut -c30-120); $(grep 'hipMalloc of' gpurun_out/${T}_cli_$S.err | cut -c24-80)"
done
cmp /tmp/fastore_bench/cli_1.cdata /tmp/fastore_bench/cli_0.cdata && echo archives identical
