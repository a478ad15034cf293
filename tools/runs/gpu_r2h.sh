export TMPDIR=/tmp
mkdir -p gpurun_out
( FS_TWO_WAVE=1 timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "ppmd or reproduces or deterministic" ) > gpurun_out/r2h_tests.log 2>&1
tail -3 gpurun_out/r2h_tests.log
FS_TWO_WAVE=1 FS_LIB=build/libfastore_amd_prof.so COPIES=1 timeout 600 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/r2h_prof_3M.txt 2>&1
cat gpurun_out/r2h_prof_3M.txt
FS_TWO_WAVE=1 COPIES=1,1536,3072 timeout 600 python3 tools/ppmd_microbench.py 1000000 > gpurun_out/r2h_micro_1M_two.txt 2>&1
cat gpurun_out/r2h_micro_1M_two.txt
( time FS_TRACE=1 FS_WATCHDOG=120 timeout 1500 python3 bench.py --steps 3 --warmup 1 --no-cli --no-cpu-baseline ) > gpurun_out/r2h_bench.json 2> gpurun_out/r2h_bench.err
cat gpurun_out/r2h_bench.json
grep "slice" gpurun_out/r2h_bench.err | tail -8
