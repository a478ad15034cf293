export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3n
# LDS per workgroup under 12 800 bytes (this build) against the commit before it (build/libfastore_amd_head.so, 12 812 bytes)
( timeout -k 10 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "every_kernel_form or ppmd_device or reproduces_reference" ) > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
tail -2 gpurun_out/${T}_tests.log
for L in new head new head; do
  if [ $L = head ]; then export FS_LIB=build/libfastore_amd_head.so; else unset FS_LIB; fi
  echo "== $L w1 x3072"; FS_WAVES=1 COPIES=3072 timeout -k 10 200 python3 tools/ppmd_microbench.py 3000000 2>&1 | tee -a gpurun_out/${T}_micro_${L}.txt | tail -2
  echo "== $L w2 lone 7M"; FS_WAVES=2 COPIES=1 timeout -k 10 120 python3 tools/ppmd_microbench.py 7000000 2>&1 | tee -a gpurun_out/${T}_micro_${L}.txt | tail -2
done
unset FS_LIB
( timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench.json')); print('SE', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'])"
