export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2i_tests.log 2>&1
tail -4 gpurun_out/r2i_tests.log
FS_TWO_WAVE=0 FS_LIB=build/libfastore_amd_prof.so COPIES=1 timeout 600 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/r2i_prof_3M_one.txt 2>&1
cat gpurun_out/r2i_prof_3M_one.txt
FS_TWO_WAVE=1 FS_LIB=build/libfastore_amd_prof.so COPIES=1 timeout 600 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/r2i_prof_3M_two.txt 2>&1
cat gpurun_out/r2i_prof_3M_two.txt
FS_TWO_WAVE=1 COPIES=1,1536 timeout 600 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/r2i_micro_3M_two.txt 2>&1
cat gpurun_out/r2i_micro_3M_two.txt
( time FS_TRACE=1 FS_WATCHDOG=120 timeout 1500 python3 bench.py --steps 3 --warmup 1 ) > gpurun_out/r2i_bench.json 2> gpurun_out/r2i_bench.err
cat gpurun_out/r2i_bench.json
grep "slice\|batch:" gpurun_out/r2i_bench.err | tail -12
