export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2xx
export GPU_MAX_HW_QUEUES=16
{
for A in "16 50 6 0 0" "16 50 6 100 0" "16 50 6 300 0" "16 50 6 0 30" "16 50 6 300 30" "1 0 0 300 0"; do
  ./build/hip_exit_probe $A; echo "   shell sees the end at $(date +%s%3N)"
done
} > gpurun_out/${T}_exit_probe2.txt 2>&1
cat gpurun_out/${T}_exit_probe2.txt
