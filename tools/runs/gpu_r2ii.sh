export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2ii
# where does the fastore_pack PROCESS spend its time on the genuine 10 M library (start -> exit 3.3-4.0 s against a 1.57 s warmed step)?
python3 - <<'PY'
import sys, time; sys.path.insert(0, '.')
import bench, os
os.makedirs('/tmp/fastore_bench', exist_ok=True)
import subprocess
os.makedirs(os.path.dirname(bench.GEN), exist_ok=True)
if not os.path.exists(bench.GEN): subprocess.check_call(['g++','-O2','-o',bench.GEN,'tools/gen_fastq.cpp'])
t=time.time()
b, size = bench.prepare_library('/tmp/fastore_bench', 'se10000k', 10_000_000, 150, 30_000_000, 8, min(os.cpu_count(), 32))
print('library ready in %.0f s' % (time.time()-t), b, size, flush=True)
PY
ls -la /tmp/fastore_bench | head -20
for i in 1 2 3; do
  TIMEFORMAT="process wall %R s user %U sys %S"; time FS_TRACE=1 ./fastore_amd/fastore_pack e -i/tmp/fastore_bench/se10000k.b8 -o/tmp/fastore_bench/cli_$i -r -f256 -c10 -d8 -w1024 -W1024 2> gpurun_out/${T}_cli_$i.err
  grep -v "^\[trace\] slice\|bin " gpurun_out/${T}_cli_$i.err | cut -c1-260 | tail -25
  echo ----
done
grep "slice" gpurun_out/${T}_cli_3.err | cut -c1-220
