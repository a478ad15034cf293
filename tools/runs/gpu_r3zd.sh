export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3zd
# every coder but PPMd out of line in the kernels (no vector spill left in any of them): parity, the PPMd forms, the lossless step against the build before the range coders
( timeout -k 10 400 python -m pytest tests/test_gpu.py -m gpu -x -q -k "rc_device or qvz_device or reproduces_reference or every_kernel_form" ) > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
tail -1 gpurun_out/${T}_tests.log
FS_WAVES=1 COPIES=3072 timeout -k 10 200 python3 tools/ppmd_microbench.py 3000000 2>&1 | tee gpurun_out/${T}_micro_w1.txt | head -1
FS_WAVES=2 COPIES=1 timeout -k 10 120 python3 tools/ppmd_microbench.py 7000000 2>&1 | tee gpurun_out/${T}_micro_w2.txt | head -1
for L in new old new old; do
  unset FASTORE_AMD_LIB
  if [ $L = old ]; then export FASTORE_AMD_LIB=$PWD/build/libfastore_amd_before_rc.so; fi
  ( timeout -k 10 400 python3 bench.py --steps 4 --warmup 2 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench_$L.json 2> gpurun_out/${T}_bench_$L.err || { tail -5 gpurun_out/${T}_bench_$L.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_$L.json')); print('$L: SE', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'])"
done
