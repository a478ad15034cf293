export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03
# the N > 1 code path of the last build, rehearsed with two ranks on the one device (gloo), 2 M-read libraries
( time timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --rehearse --reads 2000000 ) > gpurun_out/${T}_rehearse_2ranks_last.json 2> gpurun_out/${T}_rehearse_2ranks_last.err || { tail -20 gpurun_out/${T}_rehearse_2ranks_last.err; exit 1; }
cut -c1-1800 gpurun_out/${T}_rehearse_2ranks_last.json
