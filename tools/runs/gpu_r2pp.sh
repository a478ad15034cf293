export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2pp
# long-stream lanes on their own compute units (hipExtStreamCreateWithCUMask)?
for S in off 4,2,1 4,4,1 6,2,1 3,4,1 off; do
  if [ $S = off ]; then unset FS_CU_SPLIT; else export FS_CU_SPLIT=$S; fi
  N=$(echo $S | tr , _)
  FS_TRACE=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/${T}_$N.json 2> gpurun_out/${T}_$N.err || { tail -3 gpurun_out/${T}_$N.err; exit 1; }
  python3 - $N <<'PY'
import json, sys
N = sys.argv[1]
d = json.loads(open('gpurun_out/r2pp_%s.json' % N).read().strip().splitlines()[-1])
print('split', N, 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'fe', d['stages_ms_per_step_rank0']['frontend_ms'], flush=True)
PY
  grep "slice [1234]/14\|slice 14/14" gpurun_out/${T}_$N.err | tail -5 | cut -c1-230
done
