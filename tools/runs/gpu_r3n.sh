export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3n
# throughput regime (one-wave form, 3072 copies of a 3 M-symbol stream; two-wave form, 1536 copies): free-list heads in HBM / in LDS / in LDS + 32-bit windows
for v in oldheads main narrow; do
  L=build/libfastore_amd_$v.so; [ $v = main ] && L=fastore_amd/libfastore_amd.so
  echo "== $v"
  FS_WAVES=1 FS_LIB=$L COPIES=3072 timeout -k 10 200 python3 tools/ppmd_microbench.py 3000000 2>&1 | head -1
  FS_WAVES=2 FS_LIB=$L COPIES=1,1536 timeout -k 10 200 python3 tools/ppmd_microbench.py 3000000 2>&1 | grep copies
done > gpurun_out/${T}_throughput_ab.txt 2>&1
cat gpurun_out/${T}_throughput_ab.txt
