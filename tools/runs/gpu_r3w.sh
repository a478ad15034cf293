export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3w
# windowed range coders in kernels of their own: parity both ways, the lossless step against the build before them (alternating), the --reduced legs
( timeout -k 10 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "rc_device or reproduces_reference or every_kernel_form or qvz_device" ) > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
tail -2 gpurun_out/${T}_tests.log
for L in new old new old; do
  unset FASTORE_AMD_LIB
  if [ $L = old ]; then export FASTORE_AMD_LIB=$PWD/build/libfastore_amd_before_rc.so; fi
  ( timeout -k 10 400 python3 bench.py --steps 4 --warmup 2 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench_$L.json 2> gpurun_out/${T}_bench_$L.err || { tail -5 gpurun_out/${T}_bench_$L.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_$L.json')); print('$L: SE', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'])"
done
unset FASTORE_AMD_LIB
( FS_TRACE=1 timeout -k 10 600 python3 bench.py --quality reduced --steps 4 --warmup 1 --no-cli --no-pe ) > gpurun_out/${T}_bench_se_reduced.json 2> gpurun_out/${T}_bench_se_reduced.err || { tail -5 gpurun_out/${T}_bench_se_reduced.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_se_reduced.json')); print('reduced SE 10 M:', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'], d['cpu_baseline']['value'], d['parity'])"
grep "slice [1-3]/" gpurun_out/${T}_bench_se_reduced.err | tail -3 | cut -c1-230
( timeout -k 10 900 python3 bench.py --quality reduced --paired --reads 6000000 --steps 3 --warmup 1 --no-cli --no-pe ) > gpurun_out/${T}_bench_pe_reduced.json 2> gpurun_out/${T}_bench_pe_reduced.err || { tail -5 gpurun_out/${T}_bench_pe_reduced.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_pe_reduced.json')); print('reduced PE 6 M pairs:', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'], d['cpu_baseline']['value'], d['parity'])"
