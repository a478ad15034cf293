export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2ff
( timeout 1200 python -m pytest tests/test_gpu.py -m gpu -x -q -k "tokeniser or fresh_libraries or read_id" ) > gpurun_out/${T}_tests.log 2>&1
tail -3 gpurun_out/${T}_tests.log
python3 bench.py --steps 5 --warmup 2 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
cat gpurun_out/${T}_bench.json | cut -c1-1500
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/${T}_bench_under_rocprof.json 2> gpurun_out/${T}_stats.err
python3 tools/pmc_summary.py stats gpurun_out/${T}_stats > gpurun_out/${T}_kernel_stats.json
cat gpurun_out/${T}_kernel_stats.json | head -60
find gpurun_out/${T}_stats -name "*.csv" -size +1M -delete
