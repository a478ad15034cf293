export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3v
# which of the two changes costs the lossless step 4 %: the 32 MiB slots (FS_BIG_SLOTS=0 switches them off) or the kernels' code
for L in small old big small old big; do
  unset FASTORE_AMD_LIB FS_BIG_SLOTS
  if [ $L = old ]; then export FASTORE_AMD_LIB=$PWD/build/libfastore_amd_before_rc.so; fi
  if [ $L = small ]; then export FS_BIG_SLOTS=0; fi
  ( timeout -k 10 400 python3 bench.py --steps 4 --warmup 2 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench_$L.json 2> gpurun_out/${T}_bench_$L.err || { tail -5 gpurun_out/${T}_bench_$L.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_$L.json')); print('$L: SE', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'])"
done
