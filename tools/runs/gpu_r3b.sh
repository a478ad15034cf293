export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3b
# the new many-batches / many-restarts test and the two-wave hand-over tests; the driver form of the bench with its PE leg;
# the N > 1 code path rehearsed with two ranks on the one device (gloo)
( timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "many_batches or ppmd or long_streams" ) > gpurun_out/${T}_tests.log 2>&1
tail -3 gpurun_out/${T}_tests.log
( time timeout 900 python3 bench.py --steps 5 --warmup 2 ) > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || { tail -5 gpurun_out/${T}_bench.err; exit 1; }
tail -4 gpurun_out/${T}_bench.err
cut -c1-3000 gpurun_out/${T}_bench.json
( time timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --rehearse --reads 2000000 ) > gpurun_out/${T}_rehearse2.json 2> gpurun_out/${T}_rehearse2.err || { tail -20 gpurun_out/${T}_rehearse2.err; exit 1; }
cut -c1-2500 gpurun_out/${T}_rehearse2.json
