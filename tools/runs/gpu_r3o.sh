export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3o
# the whole GPU suite on the round's build
( timeout -k 10 1100 python -m pytest tests/ -m gpu -x -q ) > gpurun_out/${T}_gpu_suite.log 2>&1
tail -5 gpurun_out/${T}_gpu_suite.log
