export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3q
# the heaviest bins alone on the cores first (FS_FIRST_ROUND: threads beyond it start as bins finish)
for B in 24 12 16 8 24; do
  ( FS_FIRST_ROUND=$B FS_TRACE=1 timeout -k 10 400 python3 bench.py --steps 4 --warmup 2 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench_fr$B.json 2> gpurun_out/${T}_bench_fr$B.err || { tail -5 gpurun_out/${T}_bench_fr$B.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_fr$B.json')); print('first round $B: SE', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'])"
  grep "slice [12]/" gpurun_out/${T}_bench_fr$B.err | tail -2
done
