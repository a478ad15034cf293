#!/bin/bash
# Regression for the device stall of round 1 (more than four launches in flight: a wave could lose its ticket in the
# arena-slot ring and spin for ever).  Runs 20 default steps (eight lanes, eight launches in flight per step) under
# FS_WATCHDOG, then a stress run with few arena slots, many slices and mostly tiny streams; fails on the first hit.
#   tools/stall_repro.sh <tag>     -> gpurun_out/<tag>_stall.log
tag=${1:-r02}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/${tag}_stall.log; mkdir -p gpurun_out; : > $L
FS_WATCHDOG=60 timeout 900 python3 bench.py --steps 20 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/${tag}_stall_bench.json 2> gpurun_out/${tag}_stall_bench.err
echo "20 default steps (14 slices on 14 lanes) on the genuine 10 M library: rc=$? $(python3 -c "
import json
try:
    d=json.loads(open('gpurun_out/${tag}_stall_bench.json').read()); print('MB/s', d['value'], 'launches', d['roofline']['launches'])
except Exception as e: print('no json')")" >> $L
grep -E "watchdog" gpurun_out/${tag}_stall_bench.err | head -12 | cut -c1-200 >> $L
# few slots per XCD (waves must wait for each other's arenas), 16 slices on 14 lanes, small golden libraries: late
# workgroups of drained queues claim and release slots in quick succession
FS_WATCHDOG=60 timeout 600 python3 - >> $L 2>&1 <<'PY'
import os, sys, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import fastore_amd
from conftest import GOLDEN, manifest, knobs_from_flags
name, paired, flags = manifest()[0]
ref = open(os.path.join(GOLDEN, name + ".ref.cdata"), "rb").read()
bad = 0
with tempfile.TemporaryDirectory() as t:
    for waves in (9, 17, 40, 130):
        with fastore_amd.Packer(device_id=0, max_waves=waves, pipeline_slices=16, pipeline_lanes=14, **knobs_from_flags(flags)) as p:
            for rep in range(25):
                p.pack_file(os.path.join(GOLDEN, name + ".in"), os.path.join(t, "o"))
                bad += open(os.path.join(t, "o.cdata"), "rb").read() != ref
print("stress: 4 x 25 packs with 9..130 arena slots, 16 slices on 14 lanes: %d wrong archives, no stall" % bad)
PY
echo "stress rc=$?" >> $L
cat $L
