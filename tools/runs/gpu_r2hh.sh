export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2hh
# the N > 1 bench path, rehearsed with two ranks on the one device over gloo (set of 2 libraries of 2 M reads each)
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --reads 2000000 --rehearse > gpurun_out/${T}_set2.json 2> gpurun_out/${T}_set2.err
tail -c 1800 gpurun_out/${T}_set2.json; echo
tail -5 gpurun_out/${T}_set2.err | cut -c1-300
# the archives of the set equal the single-GPU archives of the same libraries
python3 - <<'PY'
import os, sys
sys.path.insert(0, '.')
import fastore_amd
w = '/tmp/fastore_bench'
outs = sorted(f for f in os.listdir(w) if f.startswith('out_') and f.endswith('.cdata'))
print(outs)
kn = dict(min_bin_size=256, max_lz_window=1024, max_pair_lz_window=1024, extra_reduce_hard_reads=1, min_consensus_size=10, max_hamming_distance=8)
names = ['se2000k', 'se2000k_s9']
with fastore_amd.Packer(device_id=0, **kn) as p:
    for i, n in enumerate(names):
        p.pack_file(os.path.join(w, n + '.b8'), os.path.join(w, 'single%d' % i))
        got = [o for o in outs if o.endswith('_l%d.cdata' % i)]
        same = bool(got) and open(os.path.join(w, got[-1]), 'rb').read() == open(os.path.join(w, 'single%d.cdata' % i), 'rb').read()
        print(n, got[-1:] , 'identical to the single-GPU archive:', same)
PY
( FS_TRACE=1 ./fastore_amd/fastore_pack e -i/tmp/fastore_bench/se2000k.b8 -o/tmp/fastore_bench/cli_t4 -r -f256 -c10 -d8 -w1024 -W1024 ) 2>&1 | grep "main:" | cut -c1-200
