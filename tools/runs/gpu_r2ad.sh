export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2ad
# does the in-situ time of the long streams follow the lone-stream time?  kernel time of the first slices, base against new
for V in base new base new; do
  if [ $V = new ]; then unset FASTORE_AMD_LIB; else export FASTORE_AMD_LIB=$PWD/build/libfastore_amd_$V.so; fi
  FS_TRACE=1 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/${T}_$V.json 2> gpurun_out/${T}_$V.err || { tail -3 gpurun_out/${T}_$V.err; exit 1; }
  python3 - $V <<'PY'
import json, sys, re
N = sys.argv[1]
d = json.loads(open('gpurun_out/r2ad_%s.json' % N).read().strip().splitlines()[-1])
k = {}
for line in open('gpurun_out/r2ad_%s.err' % N):
    m = re.search(r'slice (\d+)/14: .*device done at ([\d.]+) ms \(kernel ([\d.]+) ms\)', line)
    if m: k.setdefault(int(m.group(1)), []).append((float(m.group(3)), float(m.group(2))))
def avg(si, j): v = [x[j] for x in k[si][1:]]; return sum(v) / len(v)      # (first entry = warm-up step)
print(N, 'value', d['value'], 'ms_per_step', d['ms_per_step'], '| kernel ms of slices 1-4:', ' '.join('%.0f' % avg(s, 0) for s in (1, 2, 3, 4)), '| done at:', ' '.join('%.0f' % avg(s, 1) for s in (1, 2, 3, 4)), flush=True)
PY
done
