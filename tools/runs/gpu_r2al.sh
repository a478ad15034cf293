export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2al
{
for V in default park default park; do
  if [ $V = default ]; then unset FS_LIB; else export FS_LIB=$PWD/build/libfastore_amd_$V.so; fi
  echo "== $V"
  COPIES=1 python3 tools/ppmd_microbench.py 3000000 2>&1 | grep "copies" | cut -c1-140
  COPIES=1 python3 tools/ppmd_microbench.py 7000000 2>&1 | grep "copies" | cut -c1-140
  COPIES=1024 python3 tools/ppmd_microbench.py 300000 2>&1 | grep "copies" | cut -c1-140
done
} > gpurun_out/${T}_park.txt 2>&1
cat gpurun_out/${T}_park.txt
unset FS_LIB
for V in default park default park; do
  if [ $V = default ]; then unset FASTORE_AMD_LIB; else export FASTORE_AMD_LIB=$PWD/build/libfastore_amd_$V.so; fi
  python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/${T}_$V.json 2> gpurun_out/${T}_$V.err || { tail -3 gpurun_out/${T}_$V.err; exit 1; }
  python3 - $V <<'PY'
import json, sys
N = sys.argv[1]
d = json.loads(open('gpurun_out/r2al_%s.json' % N).read().strip().splitlines()[-1])
print(N, 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'encode_kernel_ms', d['stages_ms_per_step_rank0']['encode_kernel_ms'], flush=True)
PY
done
