export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2qq
# two coder waves per SIMD (256 VGPRs each, 2048 resident waves, 35.6 GB of arenas) against three (170 VGPRs, 3072, 53.4 GB)
for V in w3 w2; do
  if [ $V = w2 ]; then export FS_LIB=$PWD/build/libfastore_amd_w2.so; else unset FS_LIB; fi
  echo "== $V microbench: one 3 M-symbol stream alone, then 1536 copies of 300 k"
  COPIES=1 python3 tools/ppmd_microbench.py 3000000 2>&1 | grep -v "^ *$" | cut -c1-260
  COPIES=1536 python3 tools/ppmd_microbench.py 300000 2>&1 | grep "copies" | cut -c1-200
done > gpurun_out/${T}_micro.txt 2>&1
cat gpurun_out/${T}_micro.txt
for V in w3 w2 w3 w2; do
  if [ $V = w2 ]; then export FS_LIB=$PWD/build/libfastore_amd_w2.so; else unset FS_LIB; fi
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/${T}_$V.json 2> gpurun_out/${T}_$V.err || { tail -3 gpurun_out/${T}_$V.err; exit 1; }
  python3 - $V <<'PY'
import json, sys
N = sys.argv[1]
d = json.loads(open('gpurun_out/r2qq_%s.json' % N).read().strip().splitlines()[-1])
print(N, 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'fe', d['stages_ms_per_step_rank0']['frontend_ms'], 'matcher', d['other_kernels']['fs_match_reads'], flush=True)
PY
done
