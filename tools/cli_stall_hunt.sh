#!/bin/bash
# Many fastore_pack e processes one after the other, every one traced (FS_TRACE=1) and watched (FS_WATCHDOG): the traces of those that take
# more than 4 s are kept.   tools/cli_stall_hunt.sh <tag> [runs] [watchdog seconds]
set -u
tag=$1; runs=${2:-20}; wd=${3:-8}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
W=${FASTORE_BENCH_DIR:-/tmp/fastore_bench}
lib=$W/${FS_HUNT_LIB:-se10000k}.b8
out=gpurun_out/${tag}_stall_hunt.txt
: > $out
for r in $(seq 1 $runs); do
  s=$(date +%s.%N); FS_TRACE=1 FS_WATCHDOG=$wd fastore_amd/fastore_pack e -i$lib -o$W/cli_o2 -r -f256 -c10 -d8 -w1024 -W1024 2> gpurun_out/${tag}_stall_trace_$r.txt; rc=$?; e=$(date +%s.%N)
  t=$(python3 -c "print('%.2f' % ($e - $s))")
  echo "run $r: exit $rc, $t s" >> $out
  if python3 -c "import sys; sys.exit(0 if ($t > 4 or $rc != 0) else 1)"; then echo "---- trace of run $r ----" >> $out; grep -v "lane teardown\|matcher lane" gpurun_out/${tag}_stall_trace_$r.txt | cut -c1-300 | head -120 >> $out; fi
  rm -f gpurun_out/${tag}_stall_trace_$r.txt
done
cat $out
