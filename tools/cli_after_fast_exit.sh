#!/bin/bash
# Does a fastore_pack e process that starts right behind one that left WITHOUT teardown stall?  Alternating: a process without the preloaded library
# (leaves without teardown), then a traced one (FS_TRACE=1, FS_WATCHDOG=15).   tools/cli_after_fast_exit.sh <tag> [rounds]
set -u
tag=$1; rounds=${2:-6}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
W=${FASTORE_BENCH_DIR:-/tmp/fastore_bench}
lib=$W/se10000k.b8
out=gpurun_out/${tag}_after_fast_exit.txt
: > $out
for r in $(seq 1 $rounds); do
  s=$(date +%s.%N); env -u LD_PRELOAD fastore_amd/fastore_pack e -i$lib -o$W/cli_o -r -f256 -c10 -d8 -w1024 -W1024 2>/dev/null; rc=$?; e=$(date +%s.%N)
  echo "round $r: fast-exit process: exit $rc, $(python3 -c "print('%.2f' % ($e - $s))") s" >> $out
  s=$(date +%s.%N); FS_TRACE=1 FS_WATCHDOG=15 fastore_amd/fastore_pack e -i$lib -o$W/cli_o2 -r -f256 -c10 -d8 -w1024 -W1024 2> gpurun_out/${tag}_after_fast_exit_trace_$r.txt; rc=$?; e=$(date +%s.%N)
  t=$(python3 -c "print('%.2f' % ($e - $s))")
  echo "round $r: traced process behind it: exit $rc, $t s" >> $out
  if python3 -c "import sys; sys.exit(0 if $t > 4 else 1)"; then grep -v "lane teardown\|matcher lane" gpurun_out/${tag}_after_fast_exit_trace_$r.txt | cut -c1-220 | head -60 >> $out; else rm -f gpurun_out/${tag}_after_fast_exit_trace_$r.txt; fi
done
cat $out
