# round 3, closing measurements: the driver form of the bench (SE leg + PE leg), the same command's kernels under rocprofv3, the two HBM
# counter passes, the front end's stage clocks of the heaviest bins (the sort's share)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03
( time timeout -k 10 900 python3 bench.py --steps 8 --warmup 2 ) > gpurun_out/${T}_bench_final.json 2> gpurun_out/${T}_bench_final.err || { tail -5 gpurun_out/${T}_bench_final.err; exit 1; }
cut -c1-600 gpurun_out/${T}_bench_final.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli --no-pe > gpurun_out/${T}_bench_under_rocprof.json 2> gpurun_out/${T}_stats.err
python3 tools/pmc_summary.py stats gpurun_out/${T}_stats > gpurun_out/${T}_kernel_stats.json
head -c 1500 gpurun_out/${T}_kernel_stats.json; echo
find gpurun_out/${T}_stats -name "*.csv" -size +1M -delete
bash tools/pmc_passes.sh ${T}
python3 tools/hbm_traffic.py gpurun_out/${T}_pmc_FETCH_SIZE_summary.json gpurun_out/${T}_pmc_WRITE_SIZE_summary.json gpurun_out/${T}_pmc_FETCH_SIZE.json > gpurun_out/${T}_hbm_traffic.json
head -c 900 gpurun_out/${T}_hbm_traffic.json; echo
( FS_BIN_TRACE=1 timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cli --no-pe ) > /dev/null 2> gpurun_out/${T}_bin_trace.err
grep "^\[bin\]" gpurun_out/${T}_bin_trace.err | sort -t: -k2 | tail -12 > gpurun_out/${T}_front_end_stage_clocks.txt; cat gpurun_out/${T}_front_end_stage_clocks.txt | cut -c1-220
