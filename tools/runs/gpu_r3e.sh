export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3e
# three-wave form after the fix (contexts the episode creates join the list the window wave checks) + short spins + guessed start lane:
# parity first (nothing else runs if it fails), then the lone stream, many copies, phase clocks, the bench
( timeout -k 10 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "every_kernel_form or ppmd_device or tokeniser or reproduces_reference" ) > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
tail -2 gpurun_out/${T}_tests.log
for w in 2 3; do
  FS_WAVES=$w COPIES=1 timeout -k 10 120 python3 tools/ppmd_microbench.py 7000000 > gpurun_out/${T}_micro_7M_w$w.txt 2>&1 || { cat gpurun_out/${T}_micro_7M_w$w.txt; exit 1; }
  cat gpurun_out/${T}_micro_7M_w$w.txt
done
FS_WAVES=3 COPIES=1,64,512,1024 timeout -k 10 200 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/${T}_micro_3M_w3_copies.txt 2>&1 || { cat gpurun_out/${T}_micro_3M_w3_copies.txt; exit 1; }
cat gpurun_out/${T}_micro_3M_w3_copies.txt
FS_WAVES=2 COPIES=64,512,1024 timeout -k 10 200 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/${T}_micro_3M_w2_copies.txt 2>&1; cat gpurun_out/${T}_micro_3M_w2_copies.txt
FS_WAVES=3 FS_LIB=build/libfastore_amd_prof.so COPIES=1 timeout -k 10 120 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/${T}_prof_3M_w3.txt 2>&1; cat gpurun_out/${T}_prof_3M_w3.txt
( timeout -k 10 600 python3 bench.py --steps 5 --warmup 2 --no-cli ) > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || { tail -5 gpurun_out/${T}_bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench.json'))
print('SE', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'], d.get('parity'), d['roofline']['avg_launch_ms'])
p=d['pe']; print('PE', p['value'], p['ms_per_step'], p['stages_ms_per_step'], p.get('parity'), p.get('speedup_vs_cpu_baseline'))"
( FS_WAVES=2 timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench_w2.json 2> gpurun_out/${T}_bench_w2.err
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_w2.json')); print('SE two-wave', d['value'], d['ms_per_step'])"
