#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel stats of the same command, HBM PMC passes.
#   tools/profile_round.sh <tag>          (outputs under gpurun_out/<tag>_*)
set -u
tag=${1:-r01}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
# (the headline leg only: the four-leg line is the driver's own command, profiles/<tag>_bench_driver_form.json)
python3 bench.py --steps 2 --warmup 1 --no-pe > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli --no-pe --in-process > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_stats.err
python3 tools/pmc_summary.py stats gpurun_out/${tag}_stats > gpurun_out/${tag}_kernel_stats.json
cat gpurun_out/${tag}_bench.json
exec bash tools/pmc_passes.sh "$tag"
