export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3k
# device mate search (fs_match_mates): rows against the host's search, archives either way, PE bench with and without
( timeout -k 10 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "mate_search" ) > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
tail -2 gpurun_out/${T}_tests.log
for e in 1 0; do
( FS_DEVICE_MATES=$e FS_TRACE=1 timeout -k 10 400 python3 bench.py --paired --reads 6000000 --steps 3 --warmup 1 --no-cli --no-cpu-baseline ) > gpurun_out/${T}_pe_mates$e.json 2> gpurun_out/${T}_pe_mates$e.err
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_pe_mates$e.json')); print('PE device mates=$e', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'], d['other_kernels']['fs_match_reads'])"
grep "slice \|batch:\|packFiles total" gpurun_out/${T}_pe_mates$e.err | tail -17 | cut -c1-230
done
