export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3q
# XCD-aligned CU masks for the lanes that hold the longest streams (FS_XCD_SPLIT=<m|b>:<xcds>:<lanes>): bench step, 4 steps each
for v in off m:1:2 m:2:3 b:1:2 m:2:2; do
  if [ $v = off ]; then unset FS_XCD_SPLIT; else export FS_XCD_SPLIT=$v; fi
  ( FS_TRACE=1 timeout -k 10 300 python3 bench.py --steps 4 --warmup 2 --no-cli --no-pe --no-cpu-baseline ) > gpurun_out/${T}_$v.json 2> gpurun_out/${T}_$v.err
  python3 -c "
import json; d=json.load(open('gpurun_out/${T}_$v.json')); print('split $v:', d['value'], 'MB/s', d['ms_per_step'], 'ms')"
  grep "slice 1/\|slice 2/" gpurun_out/${T}_$v.err | tail -2 | cut -c1-200
done > gpurun_out/${T}_xcd_split.txt 2>&1
cat gpurun_out/${T}_xcd_split.txt
