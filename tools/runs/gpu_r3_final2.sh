# round 3, closing measurements on the last build: the driver form of the bench (SE leg + PE leg + --reduced leg), the lossless and the
# --reduced command's kernels under rocprofv3
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03
( time timeout -k 10 1000 python3 bench.py --steps 8 --warmup 2 ) > gpurun_out/${T}_bench_final.json 2> gpurun_out/${T}_bench_final.err || { tail -5 gpurun_out/${T}_bench_final.err; exit 1; }
cut -c1-400 gpurun_out/${T}_bench_final.json; tail -4 gpurun_out/${T}_bench_final.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli --no-pe > gpurun_out/${T}_bench_under_rocprof.json 2> gpurun_out/${T}_stats.err
python3 tools/pmc_summary.py stats gpurun_out/${T}_stats > gpurun_out/${T}_kernel_stats.json
head -c 700 gpurun_out/${T}_kernel_stats.json; echo
find gpurun_out/${T}_stats -name "*.csv" -size +1M -delete
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats_reduced -- python3 bench.py --quality reduced --steps 2 --warmup 1 --no-cpu-baseline --no-cli --no-pe > gpurun_out/${T}_bench_reduced_under_rocprof.json 2> gpurun_out/${T}_stats_reduced.err
python3 tools/pmc_summary.py stats gpurun_out/${T}_stats_reduced > gpurun_out/${T}_kernel_stats_reduced.json
head -c 700 gpurun_out/${T}_kernel_stats_reduced.json; echo
find gpurun_out/${T}_stats_reduced -name "*.csv" -size +1M -delete
