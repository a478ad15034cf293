export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "ppmd or reproduces or deterministic or sharded or cli" ) > gpurun_out/r2f_tests.log 2>&1
tail -3 gpurun_out/r2f_tests.log
FS_LIB=build/libfastore_amd_prof.so COPIES=1 timeout 600 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/r2f_prof_3M.txt 2>&1
cat gpurun_out/r2f_prof_3M.txt
( time FS_TRACE=1 FS_WATCHDOG=120 timeout 1500 python3 bench.py --steps 3 --warmup 1 --no-cli ) > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err
cat gpurun_out/r2f_bench.json
grep "slice\|batch:" gpurun_out/r2f_bench.err | tail -10
