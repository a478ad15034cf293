export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2rr
# staged read-ahead in the window step (next window's chain + the serial symbol's context chain), two waves per SIMD
for V in w2 w2pf; do
  export FS_LIB=$PWD/build/libfastore_amd_$V.so
  echo "== $V microbench: one 3 M-symbol stream alone; one 7 M; 1024 copies of 300 k"
  COPIES=1 python3 tools/ppmd_microbench.py 3000000 2>&1 | grep "copies" | cut -c1-200
  COPIES=1 python3 tools/ppmd_microbench.py 7000000 2>&1 | grep "copies" | cut -c1-200
  COPIES=1024 python3 tools/ppmd_microbench.py 300000 2>&1 | grep "copies" | cut -c1-200
done > gpurun_out/${T}_micro.txt 2>&1
cat gpurun_out/${T}_micro.txt
unset FS_LIB
for V in w2 w2pf w2 w2pf; do
  export FASTORE_AMD_LIB=$PWD/build/libfastore_amd_$V.so
  FS_TRACE=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/${T}_$V.json 2> gpurun_out/${T}_$V.err || { tail -3 gpurun_out/${T}_$V.err; exit 1; }
  python3 - $V <<'PY'
import json, sys
N = sys.argv[1]
d = json.loads(open('gpurun_out/r2rr_%s.json' % N).read().strip().splitlines()[-1])
print(N, 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'fe', d['stages_ms_per_step_rank0']['frontend_ms'], flush=True)
PY
  grep "slice [12]/14" gpurun_out/${T}_$V.err | tail -2 | cut -c1-230
done
