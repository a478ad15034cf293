cd $GRAFT_REPO_ROOT
L=gpurun_out/exp27.log; : > $L
FS_SOLO=8 FS_SOLO_MIN=1000 FS_TRACE=1 timeout 600 python3 -m pytest tests -m gpu -x -q -k "reproduces_reference or deterministic or seam" > gpurun_out/exp27_pytest.log 2>&1; echo "pytest(solo on) exit $?" >> $L; tail -2 gpurun_out/exp27_pytest.log >> $L
grep -c "solo kernel" gpurun_out/exp27_pytest.log >> $L
timeout 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
run() { echo "== $1" >> $L; shift
  FS_TRACE=1 timeout 200 env "$@" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/exp27.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('MB/s', d['value'], 'ms/step', d['ms_per_step'], d['stages_ms_per_step'])" >> $L
  grep -E "solo kernel" gpurun_out/exp27.err | tail -2 | cut -c1-170 >> $L
}
run "base" A=1
run "solo 32, 7 slices" FS_SOLO=32 FS_PIPELINE_SLICES=7
run "solo 64, 7 slices" FS_SOLO=64 FS_PIPELINE_SLICES=7
run "solo 64, 8 slices, 9 queues" FS_SOLO=64 GPU_MAX_HW_QUEUES=9
run "solo 128 (>=100k), 7 slices" FS_SOLO=128 FS_PIPELINE_SLICES=7 FS_SOLO_MIN=100000
run "7 slices, no solo" FS_PIPELINE_SLICES=7
run "base again" A=1
cat $L
