export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2ac
# short form of the rescale inside the rounds (no sorting network when the order cannot change)
{
for V in base new base new; do
  if [ $V = new ]; then unset FS_LIB; else export FS_LIB=$PWD/build/libfastore_amd_$V.so; fi
  echo "== $V"
  COPIES=1 python3 tools/ppmd_microbench.py 3000000 2>&1 | grep "copies" | cut -c1-200
  COPIES=1 python3 tools/ppmd_microbench.py 7000000 2>&1 | grep "copies" | cut -c1-200
done
export FS_LIB=$PWD/build/libfastore_amd_prof.so
echo "== phases (profile build of the new code)"
COPIES=1 python3 tools/ppmd_microbench.py 3000000 2>&1 | cut -c1-300
} > gpurun_out/${T}_micro.txt 2>&1
cat gpurun_out/${T}_micro.txt
unset FS_LIB
for V in base new base new; do
  if [ $V = new ]; then unset FASTORE_AMD_LIB; else export FASTORE_AMD_LIB=$PWD/build/libfastore_amd_$V.so; fi
  python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/${T}_$V.json 2> gpurun_out/${T}_$V.err || { tail -3 gpurun_out/${T}_$V.err; exit 1; }
  python3 - $V <<'PY'
import json, sys
N = sys.argv[1]
d = json.loads(open('gpurun_out/r2ac_%s.json' % N).read().strip().splitlines()[-1])
print(N, 'value', d['value'], 'ms_per_step', d['ms_per_step'], flush=True)
PY
done
