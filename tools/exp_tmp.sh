cd $GRAFT_REPO_ROOT
timeout 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
run() { # label, env...
  echo "== $1" >> gpurun_out/exp10.log; shift
  env "$@" FS_TRACE=1 timeout 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/exp10.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('MB/s', d['value'], 'ms/step', d['ms_per_step'], d['stages_ms_per_step'])" >> gpurun_out/exp10.log
  grep -E "slice ./|batch:" gpurun_out/exp10.err | tail -6 | cut -c1-200 >> gpurun_out/exp10.log
}
run "default build, default queues" A=1
run "default build, GPU_MAX_HW_QUEUES=8" GPU_MAX_HW_QUEUES=8
run "default build, 4 slices" FS_PIPELINE_SLICES=4
run "default build, queues 8, slices 8" GPU_MAX_HW_QUEUES=8 FS_PIPELINE_SLICES=8
run "nost48" FASTORE_AMD_LIB=$PWD/build/libfastore_nost48.so GPU_MAX_HW_QUEUES=8
run "st1" FASTORE_AMD_LIB=$PWD/build/libfastore_st1.so GPU_MAX_HW_QUEUES=8
for lib in default build/libfastore_nost48.so build/libfastore_st1.so; do
  if [ "$lib" = default ]; then unset FS_LIB; else export FS_LIB=$PWD/$lib; fi
  echo "== microbench $lib" >> gpurun_out/exp10.log
  COPIES=1,3072 timeout 100 python3 tools/ppmd_microbench.py 100000 >> gpurun_out/exp10.log 2>&1
done
cat gpurun_out/exp10.log
