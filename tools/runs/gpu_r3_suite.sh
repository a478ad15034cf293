export TMPDIR=/tmp
mkdir -p gpurun_out
# the whole GPU suite on the round's last build
( timeout -k 10 1100 python -m pytest tests/ -m gpu -x -q ) > gpurun_out/r03_gpu_suite_last.log 2>&1
echo "exit $?"; tail -6 gpurun_out/r03_gpu_suite_last.log | cut -c1-300
