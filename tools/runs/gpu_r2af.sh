export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2af
# more streams for the window searches of the first round (the coder lanes are idle then), fewer coder lanes to stay on 16 queues
run() {
  FS_TRACE=1 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/${T}_$1.json 2> gpurun_out/${T}_$1.err || { tail -3 gpurun_out/${T}_$1.err; exit 1; }
  python3 - $1 <<'PY'
import json, sys, re
N = sys.argv[1]
d = json.loads(open('gpurun_out/r2af_%s.json' % N).read().strip().splitlines()[-1])
k = {}
for line in open('gpurun_out/r2af_%s.err' % N):
    m = re.search(r'slice (\d+)/(\d+): (\d+) bins, front end done at ([\d.]+) ms.*device done at ([\d.]+) ms \(kernel ([\d.]+) ms\)', line)
    if m: k.setdefault(int(m.group(1)), []).append((int(m.group(3)), float(m.group(4)), float(m.group(5))))
def avg(si, j): v = [x[j] for x in k[si][1:]]; return sum(v) / len(v)
print(N, 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'matcher', d['other_kernels']['fs_match_reads']['host_wait_ms_per_step_summed_over_threads'], '| slices (bins ready->done):', ' '.join('%d:%.0f->%.0f' % (k[s][0][0], avg(s, 1), avg(s, 2)) for s in sorted(k)[:5]), flush=True)
PY
}
unset FS_MATCHER_STREAMS FS_PIPELINE_LANES; run m2_l14
export FS_MATCHER_STREAMS=4 FS_PIPELINE_LANES=12; run m4_l12
export FS_MATCHER_STREAMS=4 FS_PIPELINE_LANES=14; run m4_l14
export FS_MATCHER_STREAMS=6 FS_PIPELINE_LANES=10; run m6_l10
unset FS_MATCHER_STREAMS FS_PIPELINE_LANES; run m2_l14b
export FS_MATCHER_STREAMS=4 FS_PIPELINE_LANES=12; run m4_l12b
