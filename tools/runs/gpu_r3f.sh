export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3f
# three-wave form with the reply in front of the write-back; PE step traced, block 0 early or late
( timeout -k 10 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "every_kernel_form or ppmd_device" ) > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
tail -2 gpurun_out/${T}_tests.log
for w in 2 3; do
  FS_WAVES=$w COPIES=1 timeout -k 10 120 python3 tools/ppmd_microbench.py 7000000 > gpurun_out/${T}_micro_7M_w$w.txt 2>&1 || { cat gpurun_out/${T}_micro_7M_w$w.txt; exit 1; }
  cat gpurun_out/${T}_micro_7M_w$w.txt
done
FS_WAVES=3 FS_LIB=build/libfastore_amd_prof.so COPIES=1 timeout -k 10 120 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/${T}_prof_3M_w3.txt 2>&1; cat gpurun_out/${T}_prof_3M_w3.txt
for w in 3 2; do
( FS_WAVES=$w timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench_w$w.json 2> gpurun_out/${T}_bench_w$w.err
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_w$w.json')); print('SE waves $w', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'])"
done
for e in 1 0; do
( FS_BLOCK0_EARLY=$e FS_TRACE=1 timeout -k 10 400 python3 bench.py --paired --reads 6000000 --steps 3 --warmup 1 --no-cli --no-cpu-baseline ) > gpurun_out/${T}_pe_b0early$e.json 2> gpurun_out/${T}_pe_b0early$e.err
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_pe_b0early$e.json')); print('PE block0 early=$e', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'])"
grep "slice \|batch:\|packFiles total\|before final\|route" gpurun_out/${T}_pe_b0early$e.err | tail -22 | cut -c1-230
done
