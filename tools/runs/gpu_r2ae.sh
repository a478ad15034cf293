export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2ae
# finer slices at the top of the bin ranking: the second-round bins reach the device as they finish
run() {
  FS_TRACE=1 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/${T}_$1.json 2> gpurun_out/${T}_$1.err || { tail -3 gpurun_out/${T}_$1.err; exit 1; }
  python3 - $1 <<'PY'
import json, sys, re
N = sys.argv[1]
d = json.loads(open('gpurun_out/r2ae_%s.json' % N).read().strip().splitlines()[-1])
k = {}
for line in open('gpurun_out/r2ae_%s.err' % N):
    m = re.search(r'slice (\d+)/(\d+): (\d+) bins, front end done at ([\d.]+) ms.*device done at ([\d.]+) ms \(kernel ([\d.]+) ms\)', line)
    if m: k.setdefault(int(m.group(1)), []).append((int(m.group(3)), float(m.group(4)), float(m.group(5))))
def avg(si, j): v = [x[j] for x in k[si][1:]]; return sum(v) / len(v)
print(N, 'value', d['value'], 'ms_per_step', d['ms_per_step'], '| slices (bins ready->done):', ' '.join('%d:%.0f->%.0f' % (k[s][0][0], avg(s, 1), avg(s, 2)) for s in sorted(k)[:8]), flush=True)
PY
}
B="5.5,5.4,4.5,4.4,6,8,10,12,12,11,9,7,3.5,1.7"
E="5.5,5.4,4.5,4.4,5,6,8,10,12,12,11,9,5,2.2"
export FS_PIPELINE_SLICES=14 FS_SLICE_WEIGHTS=$B; run B1
unset FS_SLICE_WEIGHTS FS_PIPELINE_SLICES; run default1
export FS_PIPELINE_SLICES=14 FS_SLICE_WEIGHTS=$E; run E1
export FS_PIPELINE_SLICES=14 FS_SLICE_WEIGHTS=$B; run B2
unset FS_SLICE_WEIGHTS FS_PIPELINE_SLICES; run default2
export FS_PIPELINE_SLICES=14 FS_SLICE_WEIGHTS=$E; run E2
