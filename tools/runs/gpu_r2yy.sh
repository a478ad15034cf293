export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2yy
# first round of the host threads: one bin per core (default) against one per thread (FS_FIRST_ROUND=24, as before)
for F in 24 0 24 0 24 0; do
  if [ $F = 0 ]; then unset FS_FIRST_ROUND; else export FS_FIRST_ROUND=$F; fi
  FS_TRACE=1 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/${T}_$F.json 2> gpurun_out/${T}_$F.err || { tail -3 gpurun_out/${T}_$F.err; exit 1; }
  python3 - $F <<'PY'
import json, sys
N = sys.argv[1]
d = json.loads(open('gpurun_out/r2yy_%s.json' % N).read().strip().splitlines()[-1])
print('first_round', N, 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'fe', d['stages_ms_per_step_rank0']['frontend_ms'], flush=True)
PY
  grep "slice [12]/14" gpurun_out/${T}_$F.err | tail -2 | cut -c1-200
done
