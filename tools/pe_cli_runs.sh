#!/bin/bash
# usage: pe_runs.sh "<env assignments>" ...   -> wall times of the PE 25M CLI under each
W=${FASTORE_BENCH_DIR:-/tmp/fastore_bench}
lib=$W/pe25000k.b8
for e in "$@"; do
  for i in 1 2; do
    s=$(date +%s.%N); env $e fastore_amd/fastore_pack e -i$lib -o$W/cli_pe_o -r -f256 -c10 -d8 -w1024 -W1024 -z 2>/dev/null; rc=$?; t=$(date +%s.%N)
    echo "$e run $i: exit $rc $(python3 -c "print('%.2f' % ($t - $s))") s"
  done
done
