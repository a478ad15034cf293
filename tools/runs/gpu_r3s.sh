export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3s
# configs[3]'s mode (--reduced: 8-bin scores, range-coded) and --lossy (QVZ) on one device: one library each, whole leg (step, reference -t32, parity)
for Q in reduced lossy; do
  ( timeout -k 10 900 python3 bench.py --quality $Q --steps 3 --warmup 1 --no-cli --no-pe ) > gpurun_out/${T}_bench_se_$Q.json 2> gpurun_out/${T}_bench_se_$Q.err || { tail -5 gpurun_out/${T}_bench_se_$Q.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_se_$Q.json')); print('$Q SE 10 M:', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'], d['cpu_baseline'], d['parity'])"
done
( timeout -k 10 1100 python3 bench.py --quality reduced --paired --reads 6000000 --steps 3 --warmup 1 --no-cli --no-pe ) > gpurun_out/${T}_bench_pe_reduced.json 2> gpurun_out/${T}_bench_pe_reduced.err || { tail -5 gpurun_out/${T}_bench_pe_reduced.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_pe_reduced.json')); print('reduced PE 6 M pairs:', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'], d['cpu_baseline'], d['parity'])"
