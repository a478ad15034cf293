export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3zb
# the windowed kernels chosen by the share of range-coded symbols: lossless (old kernels again) against the build before the range coders, the --reduced leg, the rehearsal
( timeout -k 10 300 python -m pytest tests/test_gpu.py -m gpu -x -q -k "rc_device or reproduces_reference" ) > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
tail -1 gpurun_out/${T}_tests.log
for L in new old new old; do
  unset FASTORE_AMD_LIB
  if [ $L = old ]; then export FASTORE_AMD_LIB=$PWD/build/libfastore_amd_before_rc.so; fi
  ( timeout -k 10 400 python3 bench.py --steps 4 --warmup 2 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench_$L.json 2> gpurun_out/${T}_bench_$L.err || { tail -5 gpurun_out/${T}_bench_$L.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_$L.json')); print('$L: SE', d['value'], d['ms_per_step'], d['stages_ms_per_step_rank0'])"
done
unset FASTORE_AMD_LIB
( timeout -k 10 600 python3 bench.py --quality reduced --steps 3 --warmup 1 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_bench_se_reduced.json 2> gpurun_out/${T}_bench_se_reduced.err || { tail -5 gpurun_out/${T}_bench_se_reduced.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_bench_se_reduced.json')); print('reduced SE 10 M:', d['value'], d['ms_per_step'])"
( timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --rehearse --reads 2000000 ) > gpurun_out/${T}_rehearse2.json 2> gpurun_out/${T}_rehearse2.err || { tail -20 gpurun_out/${T}_rehearse2.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_rehearse2.json')); print('rehearsal 2 ranks:', d['value'], d['ms_per_step'], d['strong']['value'])"
