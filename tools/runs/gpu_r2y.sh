export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2y
( FS_BIN_TRACE=1 FS_TRACE=1 timeout 600 python3 bench.py --steps 2 --warmup 1 --no-cli --no-cpu-baseline ) > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
grep "^\[bin\]" gpurun_out/${T}_bench.err | tail -24 | cut -c1-220
grep "slice 1/\|slice 2/\|batch:" gpurun_out/${T}_bench.err | tail -3 | cut -c1-170
