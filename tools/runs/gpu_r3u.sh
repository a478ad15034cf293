export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3u
# the lossless step on the build before the windowed range coder / big slots (build/libfastore_amd_before_rc.so) and on this one, alternating on one box
for L in new old new old new old; do
  if [ $L = old ]; then export FASTORE_AMD_LIB=$PWD/build/libfastore_amd_before_rc.so; else unset FASTORE_AMD_LIB; fi
  ( timeout -k 10 400 python3 bench.py --steps 4 --warmup 2 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench_$L.json 2> gpurun_out/${T}_bench_$L.err || { tail -5 gpurun_out/${T}_bench_$L.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_$L.json')); print('$L: SE', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'])"
done
