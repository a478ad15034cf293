cd $GRAFT_REPO_ROOT
build/store_ack_probe > gpurun_out/probe.log 2>&1; cat gpurun_out/probe.log
python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
for s in 1 2 3; do for w in 0 2048; do
  echo "== slices $s max_waves $w" >> gpurun_out/exp6.log
  FS_PIPELINE_SLICES=$s FS_MAX_WAVES=$w FS_TRACE=1 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/exp6.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('MB/s', d['value'], 'ms/step', d['ms_per_step'], d['stages_ms_per_step'])" >> gpurun_out/exp6.log
  grep -E "slice . done|batch:" gpurun_out/exp6.err | tail -4 >> gpurun_out/exp6.log
done; done
cat gpurun_out/exp6.log
