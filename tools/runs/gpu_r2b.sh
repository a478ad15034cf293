export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "ppmd or reproduces or deterministic" ) > gpurun_out/r2b_tests.log 2>&1
tail -5 gpurun_out/r2b_tests.log
COPIES=1 timeout 600 python3 tools/ppmd_microbench.py 7000000 > gpurun_out/r2b_micro_7M.txt 2>&1
cat gpurun_out/r2b_micro_7M.txt
COPIES=1,1024,3072,6144 timeout 600 python3 tools/ppmd_microbench.py 1000000 > gpurun_out/r2b_micro_1M.txt 2>&1
cat gpurun_out/r2b_micro_1M.txt
( time timeout 1200 python3 bench.py --steps 2 --warmup 1 --no-cli ) > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err
cat gpurun_out/r2b_bench.json
