export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2ag
{
for V in default nounroll os default nounroll os; do
  if [ $V = default ]; then unset FS_LIB; else export FS_LIB=$PWD/build/libfastore_amd_$V.so; fi
  echo "== $V"
  COPIES=1 python3 tools/ppmd_microbench.py 3000000 2>&1 | grep "copies" | cut -c1-140
  COPIES=1024 python3 tools/ppmd_microbench.py 300000 2>&1 | grep "copies" | cut -c1-140
done
} > gpurun_out/${T}_flags.txt 2>&1
cat gpurun_out/${T}_flags.txt
