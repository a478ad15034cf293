# BASELINE configs[2] (100 M pairs x 150 bp PE, --lossless) scaled to what one gpurun call (20 minutes, prep included) holds:
#   PAIRS=40000000 bash tools/runs/gpu_r3_config2.sh      -> gpurun_out/r3_config2_*.json
# The library is generated and binned by the real reference tools in /dev/shm (RAM-backed: the box's disk is 79 GB), packed by
# the product and by the reference at -t32, every block compared (bench.py: parity).
export TMPDIR=/tmp
mkdir -p gpurun_out
PAIRS=${PAIRS:-40000000}
T=r3_config2_$((PAIRS/1000000))M
( while true; do echo "$(date +%T) $(du -sh /dev/shm/fb 2>/dev/null | cut -f1) $(free -g | awk '/Mem:/{print $3" GB used"}')" >> gpurun_out/${T}_heartbeat.txt; sleep 45; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cat /sys/fs/cgroup/memory.max /sys/fs/cgroup/cpu.max 2>/dev/null > gpurun_out/${T}_cgroup.txt
( time FS_TRACE=1 timeout -k 10 1100 python3 bench.py --paired --reads $PAIRS --steps 2 --warmup 1 --no-cli --work /dev/shm/fb ) > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
rc=$?
grep -v "^\[trace\]" gpurun_out/${T}_bench.err | tail -8
grep "packFiles total\|batch:" gpurun_out/${T}_bench.err | tail -12 | cut -c1-200
cut -c1-2500 gpurun_out/${T}_bench.json
rm -rf /dev/shm/fb
exit $rc
