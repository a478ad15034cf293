export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2xx
export GPU_MAX_HW_QUEUES=16
{
for A in "1 0 0" "16 0 0" "16 50 0" "16 0 6" "16 50 6" "1 50 6" "4 50 6"; do
  ./build/hip_exit_probe $A; echo "   shell sees the end at $(date +%s%3N)"
done
} > gpurun_out/${T}_exit_probe.txt 2>&1
cat gpurun_out/${T}_exit_probe.txt
