export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2jj
P=./build/hip_startup_probe
TIMEFORMAT="   process wall %R s"
for m in "0 0" "0 1" "0 2" "1 0" "2 0" "3 0" "4 0"; do
  echo "== mode $m (alloc-mode, exit: 0 return / 1 frees / 2 _exit)"; time $P $m
done > gpurun_out/${T}_probe.txt 2>&1
cat gpurun_out/${T}_probe.txt
