export TMPDIR=/tmp
mkdir -p gpurun_out
python3 bench.py --steps 1 --warmup 0 --no-cli --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY' > gpurun_out/r2q_matcher_alone.txt 2>&1
import sys, time
sys.path.insert(0, '.')
import fastore_amd
kn = dict(min_bin_size=256, max_lz_window=1024, max_pair_lz_window=1024, extra_reduce_hard_reads=1, min_consensus_size=10, max_hamming_distance=8)
with fastore_amd.Packer(device_id=0, **kn) as p:
    for rep in range(2):
        p.reset_stats(); t = time.time()
        n, d = p.matcher_check('/tmp/fastore_bench/se10000k.b8')
        st = p.stats()
        print("matcher alone: %d reads, %d differing, wall %.2f s, calls %.0f ms (summed), kernels %.0f ms (summed) -> %.3f us of kernel time per read" % (n, d, time.time() - t, st["matcher_call_ms"], st["matcher_kernel_ms"], st["matcher_kernel_ms"] * 1e3 / max(1, st["matcher_reads"])), flush=True)
PY
cat gpurun_out/r2q_matcher_alone.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2q_stats -- python3 -c "
import sys
sys.path.insert(0, '.')
import fastore_amd
kn = dict(min_bin_size=256, max_lz_window=1024, max_pair_lz_window=1024, extra_reduce_hard_reads=1, min_consensus_size=10, max_hamming_distance=8)
with fastore_amd.Packer(device_id=0, **kn) as p:
    print(p.matcher_check('/tmp/fastore_bench/se10000k.b8'))
" > gpurun_out/r2q_rocprof.log 2>&1
find gpurun_out/r2q_stats -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200 | head -12
find gpurun_out/r2q_stats -name "*.csv" -size +2M -delete
