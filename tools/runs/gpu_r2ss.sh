export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2ss
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.txt 2>&1; rc=$?
tail -5 gpurun_out/${T}_pytest.txt
[ $rc = 0 ] || exit $rc
