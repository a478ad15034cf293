export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "ppmd" ) > gpurun_out/r2e_tests.log 2>&1
tail -3 gpurun_out/r2e_tests.log
FS_LIB=build/libfastore_amd_prof.so COPIES=1 timeout 600 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/r2e_prof_3M.txt 2>&1
cat gpurun_out/r2e_prof_3M.txt
COPIES=1,3072 timeout 600 python3 tools/ppmd_microbench.py 1000000 > gpurun_out/r2e_micro_1M.txt 2>&1
cat gpurun_out/r2e_micro_1M.txt
