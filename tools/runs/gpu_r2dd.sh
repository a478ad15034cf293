export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2dd
( timeout 1200 python -m pytest tests/test_gpu.py -m gpu -x -q -k "gather or quality or reproduces or fresh_library or cli or seam or flags" ) > gpurun_out/${T}_tests.log 2>&1
tail -4 gpurun_out/${T}_tests.log
