export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3h
# three-wave form: mailboxes read with one LDS access, forecast watched inside the solve
( timeout -k 10 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "every_kernel_form or ppmd_device" ) > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
tail -2 gpurun_out/${T}_tests.log
for w in 2 3; do
  FS_WAVES=$w COPIES=1 timeout -k 10 120 python3 tools/ppmd_microbench.py 7000000 > gpurun_out/${T}_micro_7M_w$w.txt 2>&1 || { cat gpurun_out/${T}_micro_7M_w$w.txt; exit 1; }
  cat gpurun_out/${T}_micro_7M_w$w.txt
done
FS_WAVES=3 FS_LIB=build/libfastore_amd_prof.so COPIES=1 timeout -k 10 120 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/${T}_scoutprof_3M_w3.txt 2>&1; cat gpurun_out/${T}_scoutprof_3M_w3.txt
