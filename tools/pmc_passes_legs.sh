#!/bin/bash
# HBM traffic counters of the legs beside the headline (separate rocprofv3 --pmc passes, nothing else traced; one pass = ONE pack of the leg's
# library in the profiled process), and each leg's kernel statistics.   tools/pmc_passes_legs.sh <tag> [pe-pairs] [legs...]
#   -> gpurun_out/<tag>_pmc_<leg>_<COUNTER>_summary.json, gpurun_out/<tag>_hbm_traffic_<leg>.json, gpurun_out/<tag>_kernel_stats_<leg>.json
set -u
tag=${1:-r01}; pairs=${2:-25000000}; shift; shift
legs=${*:-reduced lossy pe}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for leg in $legs; do
  case $leg in
    reduced) args="--quality reduced";;
    lossy) args="--quality lossy";;
    pe) args="--paired --reads $pairs";;
  esac
  common="$args --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-pe --in-process"
  python3 bench.py $common > gpurun_out/${tag}_bench_${leg}.json 2> gpurun_out/${tag}_bench_${leg}.err      # (prepares the library; the line the traffic is set against)
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d gpurun_out/${tag}_pmc_${leg}_$c -- python3 bench.py $common > gpurun_out/${tag}_pmc_${leg}_$c.json 2> gpurun_out/${tag}_pmc_${leg}_$c.err
    python3 tools/pmc_summary.py pmc gpurun_out/${tag}_pmc_${leg}_$c > gpurun_out/${tag}_pmc_${leg}_${c}_summary.json
    rm -rf gpurun_out/${tag}_pmc_${leg}_$c
  done
  python3 tools/hbm_traffic.py gpurun_out/${tag}_pmc_${leg}_FETCH_SIZE_summary.json gpurun_out/${tag}_pmc_${leg}_WRITE_SIZE_summary.json gpurun_out/${tag}_bench_${leg}.json > gpurun_out/${tag}_hbm_traffic_${leg}.json
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_${leg} -- python3 bench.py $common > /dev/null 2> gpurun_out/${tag}_stats_${leg}.err
  python3 tools/pmc_summary.py stats gpurun_out/${tag}_stats_${leg} > gpurun_out/${tag}_kernel_stats_${leg}.json
  rm -rf gpurun_out/${tag}_stats_${leg}
  echo "leg $leg done: $(python3 -c "import json; d=json.load(open('gpurun_out/${tag}_hbm_traffic_${leg}.json')); print(d.get('hbm_bytes_per_step'), d.get('traffic_over_algorithmic'))")"
done
