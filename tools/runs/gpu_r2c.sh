export TMPDIR=/tmp
mkdir -p gpurun_out
FS_LIB=build/libfastore_amd_prof.so COPIES=1 timeout 600 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/r2c_prof_3M.txt 2>&1
cat gpurun_out/r2c_prof_3M.txt
FS_LIB=build/libfastore_amd_prof.so COPIES=3072 timeout 600 python3 tools/ppmd_microbench.py 300000 > gpurun_out/r2c_prof_300k.txt 2>&1
cat gpurun_out/r2c_prof_300k.txt
