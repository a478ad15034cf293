export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3a
# round 3 opening: what the box has (cores, memory, disk), the GPU suite as it stands, lone-stream baselines
( nproc; free -g; df -h /tmp /dev/shm . ; lscpu | head -20; ulimit -a ) > gpurun_out/${T}_box.txt 2>&1
cat gpurun_out/${T}_box.txt
( timeout 900 python -m pytest tests/ -m gpu -x -q ) > gpurun_out/${T}_tests.log 2>&1
tail -3 gpurun_out/${T}_tests.log
FS_TWO_WAVE=1 FS_LIB=build/libfastore_amd_prof.so COPIES=1 timeout 300 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/${T}_prof_3M_two.txt 2>&1
cat gpurun_out/${T}_prof_3M_two.txt
FS_TWO_WAVE=1 COPIES=1 timeout 300 python3 tools/ppmd_microbench.py 7000000 > gpurun_out/${T}_micro_7M_two.txt 2>&1
cat gpurun_out/${T}_micro_7M_two.txt
