cd $GRAFT_REPO_ROOT
(cat /sys/fs/cgroup/cpu.max; nproc; grep Cpus_allowed_list /proc/self/status; cat /sys/fs/cgroup/cpu.stat | head -6; cat /sys/fs/cgroup/memory.max) > gpurun_out/exp19.log 2>&1
build/cpu_scaling_probe >> gpurun_out/exp19.log 2>&1
timeout 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
run() { echo "== $1" >> gpurun_out/exp19.log; shift
  FS_TRACE=1 timeout 200 "$@" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/exp19.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('MB/s', d['value'], 'ms/step', d['ms_per_step'], d['stages_ms_per_step'])" >> gpurun_out/exp19.log
  grep -E "batch:" gpurun_out/exp19.err | tail -1 | cut -c1-220 >> gpurun_out/exp19.log
}
run "blocking waits, 24 threads" env A=1
run "blocking waits, 48 threads" env FS_HOST_THREADS=48
(cat /sys/fs/cgroup/cpu.stat | head -6) >> gpurun_out/exp19.log 2>&1
cat gpurun_out/exp19.log
