export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3p
( timeout -k 10 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "two_pipelines or many_batches" ) > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
tail -2 gpurun_out/${T}_tests.log
PAIRS=60000000 bash tools/runs/gpu_r3_config2.sh
