export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3p
# how many of the heaviest bins have their window searches on the device: the cap measured again on this round's build
for B in 144 288 576 2000 144; do
  ( FS_MATCHER_BINS=$B timeout -k 10 400 python3 bench.py --steps 4 --warmup 2 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench_bins$B.json 2> gpurun_out/${T}_bench_bins$B.err || { tail -5 gpurun_out/${T}_bench_bins$B.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_bins$B.json')); print('matcher bins $B: SE', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'], d['other_kernels']['fs_match_reads'])"
done
