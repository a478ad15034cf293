export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2aj
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
import bench, os
os.makedirs('/tmp/fastore_bench', exist_ok=True)
bench.prepare_library('/tmp/fastore_bench', 'se10000k', 10_000_000, 150, 30_000_000, 8, min(os.cpu_count(), 32))
PY
# back-to-back runs (a directory of libraries packed one after the other): every process inherits the driver's clearing of the
# previous one's device memory -- does a smaller arena pool shorten that?
for W in 3072 3072 3072 3072 3072 1536 1536 1536 1536 1536 1024 1024 1024 1024 1024; do
  A=$(date +%s%3N)
  FS_MAX_WAVES=$W FS_TRACE=1 ./fastore_amd/fastore_pack e -i/tmp/fastore_bench/se10000k.b8 -o/tmp/fastore_bench/cli_w -r -f256 -c10 -d8 -w1024 -W1024 2> gpurun_out/${T}_cli.err
  B=$(date +%s%3N)
  echo "max_waves=$W: process wall $((B-A)) ms; $(grep 'main: context' gpurun_out/${T}_cli.err | cut -c15-75); $(grep 'hipMalloc of' gpurun_out/${T}_cli.err | cut -c24-80); $(grep 'slice 1/14' gpurun_out/${T}_cli.err | cut -c30-75)"
done
