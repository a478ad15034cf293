#!/bin/bash
# HBM traffic counters of the bench command (separate rocprofv3 --pmc passes, nothing else traced).
# One pass = ONE pack of the library in the profiled process (no CLI children, no PE leg, no reference run).
#   tools/pmc_passes.sh <tag>      -> gpurun_out/<tag>_pmc_<COUNTER>_summary.json
set -u
tag=${1:-r01}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/${tag}_pmc_$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-pe --in-process > gpurun_out/${tag}_pmc_$c.json 2> gpurun_out/${tag}_pmc_$c.err
  python3 tools/pmc_summary.py pmc gpurun_out/${tag}_pmc_$c > gpurun_out/${tag}_pmc_${c}_summary.json
  find gpurun_out/${tag}_pmc_$c -name "*.csv" -size +2M -delete
done
