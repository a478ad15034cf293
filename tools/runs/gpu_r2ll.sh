export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2ll
P=./build/hip_pool_probe
TIMEFORMAT="   process wall %R s"
{
time $P d; time $P 1; time $P d; time $P m 8; time $P d; time $P v 8; time $P d; time $P m 32; time $P d; time $P v 32
} > gpurun_out/${T}_pool_probe.txt 2>&1
cat gpurun_out/${T}_pool_probe.txt
