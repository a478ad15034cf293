export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2x
( timeout 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "ppmd or reproduces or deterministic" ) > gpurun_out/${T}_tests.log 2>&1
tail -3 gpurun_out/${T}_tests.log
FS_TWO_WAVE=1 FS_LIB=build/libfastore_amd_prof.so COPIES=1 timeout 600 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/${T}_prof_3M_two.txt 2>&1
cat gpurun_out/${T}_prof_3M_two.txt
FS_TWO_WAVE=1 COPIES=1 timeout 600 python3 tools/ppmd_microbench.py 7000000 > gpurun_out/${T}_micro_7M_two.txt 2>&1
cat gpurun_out/${T}_micro_7M_two.txt
bash tools/stall_repro.sh r2x
