export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3zc
# the 2-rank rehearsal with the work-stealing tail off (the default now) and on
for S in 0 1; do
  ( FS_STEAL=$S timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2951$S bench.py --gpus 2 --steps 2 --warmup 1 --rehearse --reads 2000000 ) > gpurun_out/${T}_rehearse2_steal$S.json 2> gpurun_out/${T}_rehearse2_steal$S.err || { tail -20 gpurun_out/${T}_rehearse2_steal$S.err; exit 1; }
  grep '^{' gpurun_out/${T}_rehearse2_steal$S.json > gpurun_out/${T}_line$S.json
  python3 -c "
import json; d=json.load(open('gpurun_out/${T}_line$S.json')); print('rehearsal 2 ranks, FS_STEAL=$S:', d['value'], d['ms_per_step'], 'strong', d['strong']['value'], d['strong']['ms_per_step'])"
done
