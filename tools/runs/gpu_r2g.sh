export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2g_tests.log 2>&1
tail -4 gpurun_out/r2g_tests.log
bash tools/stall_repro.sh r2g
