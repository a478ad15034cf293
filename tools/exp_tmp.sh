cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests -m gpu -x -q -k "seam" > gpurun_out/r01f_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r01f_pytest.log
tail -3 gpurun_out/r01f_pytest.log
timeout 1200 python3 bench.py --paired --libs 5 --reads-per-lib 1000000 --steps 2 --warmup 1 > gpurun_out/r01f_bench_pe.json 2> gpurun_out/r01f_bench_pe.err
tail -3 gpurun_out/r01f_bench_pe.err
cat gpurun_out/r01f_bench_pe.json
