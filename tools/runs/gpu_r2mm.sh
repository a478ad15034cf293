export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2mm
P=./build/hip_pool_probe
{
echo "== back to back: size in GB -> first-allocation time"
for X in 50 50 50 25 25 25 12 12 12 6 6 6 50 50; do echo -n "$X GB: "; $P 1 1 $X | grep hipMalloc; done
echo "== 4 s pause before each"
for X in 50 50 50 25 25 12 12 50; do sleep 4; echo -n "$X GB: "; $P 1 1 $X | grep hipMalloc; done
echo "== 12 s pause before each"
for X in 50 50 50; do sleep 12; echo -n "$X GB: "; $P 1 1 $X | grep hipMalloc; done
} > gpurun_out/${T}_alloc_sizes.txt 2>&1
cat gpurun_out/${T}_alloc_sizes.txt
