export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2nn
python3 - <<'PY'
import sys, time; sys.path.insert(0, '.')
import bench, os
os.makedirs('/tmp/fastore_bench', exist_ok=True)
t=time.time()
b, size = bench.prepare_library('/tmp/fastore_bench', 'se10000k', 10_000_000, 150, 30_000_000, 8, min(os.cpu_count(), 32))
print('library ready in %.0f s' % (time.time()-t), b, size, flush=True)
PY
TIMEFORMAT="process wall %R s user %U sys %S"
for i in 1 2 3 4; do
  ( time FS_TRACE=1 FS_BIN_TRACE=1 ./fastore_amd/fastore_pack e -i/tmp/fastore_bench/se10000k.b8 -o/tmp/fastore_bench/cli_$i -r -f256 -c10 -d8 -w1024 -W1024 ) 2> gpurun_out/${T}_cli_$i.err
  grep -v "^\[trace\] slice\|^\[bin\]" gpurun_out/${T}_cli_$i.err | cut -c1-260 | tail -22
  echo ----
done
grep "slice" gpurun_out/${T}_cli_4.err | cut -c1-260
grep "^\[bin\]" gpurun_out/${T}_cli_4.err | head -12 | cut -c1-260
echo "==== one context, three packs (cold, then warmed)"
FS_TRACE=1 FS_BIN_TRACE=1 python3 - 2> gpurun_out/${T}_py.err <<'PY'
import sys, time; sys.path.insert(0, '.')
import fastore_amd
kn = dict(min_bin_size=256, max_lz_window=1024, max_pair_lz_window=1024, extra_reduce_hard_reads=1, min_consensus_size=10, max_hamming_distance=8)
with fastore_amd.Packer(device_id=0, **kn) as p:
    for i in range(3):
        sys.stderr.write('=== pack %d\n' % i); sys.stderr.flush()
        t = time.time(); p.pack_file('/tmp/fastore_bench/se10000k.b8', '/tmp/fastore_bench/py_%d' % i); print('pack %d: %.0f ms' % (i, 1e3 * (time.time() - t)), flush=True)
PY
grep -n "=== pack\|slice [123]/\|set-up\|batch:" gpurun_out/${T}_py.err | cut -c1-260
grep "^\[bin\]" gpurun_out/${T}_py.err | tail -6 | cut -c1-260
cmp /tmp/fastore_bench/cli_4.cdata /tmp/fastore_bench/py_2.cdata && echo archives identical
