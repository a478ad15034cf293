export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2oo
# does the step need 3072 resident coder waves (53 GB of arenas)?  The step is bound by the latency of its longest streams.
for W in 3072 2048 1536 1024 768 512; do
  FS_MAX_WAVES=$W python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/${T}_w$W.json 2> gpurun_out/${T}_w$W.err || { tail -3 gpurun_out/${T}_w$W.err; exit 1; }
  python3 - $W <<'PY'
import json, sys
W = sys.argv[1]
d = json.loads(open('gpurun_out/r2oo_w%s.json' % W).read().strip().splitlines()[-1])
print('max_waves', W, 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'stages', d['stages_ms_per_step_rank0'], 'matcher', d['other_kernels']['fs_match_reads'], flush=True)
PY
done
