export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3za
# the test the suite stopped in, alone, with the timeline on stderr
( FS_TRACE=1 FS_WATCHDOG=90 timeout -k 10 420 python -m pytest tests/test_gpu.py -m gpu -x -q -s -k "long_streams and False" ) > gpurun_out/${T}_long_streams.log 2>&1
echo "exit $?"; tail -25 gpurun_out/${T}_long_streams.log | cut -c1-300
