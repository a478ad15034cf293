export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2a_tests.log 2>&1
tail -5 gpurun_out/r2a_tests.log
( time FS_TRACE=1 timeout 1200 python3 bench.py --steps 2 --warmup 1 ) > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
cat gpurun_out/r2a_bench.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2a_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/r2a_bench_under_rocprof.json 2> gpurun_out/r2a_stats.err
python3 tools/pmc_summary.py stats gpurun_out/r2a_stats > gpurun_out/r2a_kernel_stats.json
find gpurun_out/r2a_stats -name "*.csv" -size +2M -delete
nproc; lscpu | grep "Model name"; cat /sys/fs/cgroup/cpu.max 2>/dev/null
