export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3i
( timeout -k 10 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "every_kernel_form" ) > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
tail -2 gpurun_out/${T}_tests.log
FS_WAVES=3 COPIES=1 timeout -k 10 120 python3 tools/ppmd_microbench.py 7000000 2>&1 | head -1
FS_WAVES=3 FS_LIB=build/libfastore_amd_w3wait.so COPIES=1 timeout -k 10 120 python3 tools/ppmd_microbench.py 7000000 2>&1 | head -1
FS_WAVES=3 FS_LIB=build/libfastore_amd_prof.so COPIES=1 timeout -k 10 120 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/${T}_scoutprof_wait_3M_w3.txt 2>&1; cat gpurun_out/${T}_scoutprof_wait_3M_w3.txt
