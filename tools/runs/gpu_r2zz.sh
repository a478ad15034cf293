export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2zz
# final numbers of the round: driver form of the bench (N = 1), the same command under rocprofv3, the PE side measurement
python3 bench.py --steps 5 --warmup 2 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || { tail -5 gpurun_out/${T}_bench.err; exit 1; }
cat gpurun_out/${T}_bench.json | cut -c1-1800
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/${T}_bench_under_rocprof.json 2> gpurun_out/${T}_stats.err
python3 tools/pmc_summary.py stats gpurun_out/${T}_stats > gpurun_out/${T}_kernel_stats.json
head -c 2500 gpurun_out/${T}_kernel_stats.json; echo
find gpurun_out/${T}_stats -name "*.csv" -size +1M -delete
python3 bench.py --paired --reads 5000000 --steps 2 --warmup 1 --no-cli > gpurun_out/${T}_bench_pe.json 2> gpurun_out/${T}_bench_pe.err || { tail -5 gpurun_out/${T}_bench_pe.err; exit 1; }
cat gpurun_out/${T}_bench_pe.json | cut -c1-1200
