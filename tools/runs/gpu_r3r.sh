export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3r
# block 0's host threads behind the heaviest bins' front ends (FS_BLOCK0_AFTER_BINS; 0 = at once as before)
for B in 0 72 48 120 0 72; do
  ( FS_BLOCK0_AFTER_BINS=$B FS_TRACE=1 timeout -k 10 400 python3 bench.py --steps 4 --warmup 2 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench_b0after$B.json 2> gpurun_out/${T}_bench_b0after$B.err || { tail -5 gpurun_out/${T}_bench_b0after$B.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_b0after$B.json')); print('block 0 after $B bins: SE', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'])"
  grep "slice [1-5]/" gpurun_out/${T}_bench_b0after$B.err | tail -5 | cut -c1-75,150-230
done
( FS_TRACE=1 timeout -k 10 600 python3 bench.py --paired --reads 6000000 --steps 3 --warmup 1 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench_pe.json 2> gpurun_out/${T}_bench_pe.err || { tail -5 gpurun_out/${T}_bench_pe.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_pe.json')); print('PE 6 M pairs:', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'])"
