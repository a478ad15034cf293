#!/bin/bash
# PMC passes over fs_encode_streams on the PPMd micro-benchmark (3072 copies of one 100 k-symbol quality stream = one
# full wave set).  One rocprofv3 --pmc pass per line of <passes-file>; nothing else is traced.
#   tools/pmc_microbench.sh <tag> <passes-file>
set -u
tag=${1:-r01}; passes=${2:-tools/pmc_passes_default.txt}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp COPIES=3072
mkdir -p gpurun_out
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line --output-format csv -d gpurun_out/${tag}_mb_pass$i -- python3 tools/ppmd_microbench.py 100000 > gpurun_out/${tag}_mb_pass$i.log 2> gpurun_out/${tag}_mb_pass$i.err
  python3 tools/pmc_summary.py pmc gpurun_out/${tag}_mb_pass$i > gpurun_out/${tag}_mb_pass${i}_summary.json
  rm -rf gpurun_out/${tag}_mb_pass$i
done < "$passes"
grep -h "copies" gpurun_out/${tag}_mb_pass*.log | head -3
