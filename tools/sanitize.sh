#!/bin/bash
# Host pipeline under the sanitizers (CPU only: the test-only emulation stands in for the kernels): every golden fixture through
# fastore_pack built with -fsanitize=address,undefined and with -fsanitize=thread, one context and two (-G2: the bin-sharded path) and with small device batches (two pipelines in one process); the archives must equal the reference's and the sanitizers must stay silent.
#   tools/sanitize.sh [asan|tsan]        (default: both)
set -u
cd "$(dirname "$0")/.."
mkdir -p build/asan
SRC="fastore_amd/csrc/binfile.cpp fastore_amd/csrc/frontend.cpp fastore_amd/csrc/qvz.cpp fastore_amd/csrc/packer.cpp fastore_amd/csrc/hostcoders.cpp fastore_amd/csrc/capi.cpp tests/emu/engine_emu.cpp fastore_amd/csrc/pack_main.cpp"
rc=0
for kind in ${1:-asan tsan}; do
  if [ $kind = asan ]; then FL="-fsanitize=address,undefined -fno-sanitize-recover=undefined"; else FL="-fsanitize=thread"; fi
  g++ -O1 -g -std=c++17 -pthread -mpopcnt $FL -o build/asan/fastore_pack_$kind $SRC || exit 1
  while read name paired flags; do
    for g in "" "-G2" "split"; do
      out=/tmp/san_${kind}_${name}${g}
      bb=0; gg=$g; if [ "$g" = split ]; then bb=300000; gg=""; fi      # split: small device batches -> two pipelines in one process (capi.cpp: packSplit)
      FS_BATCH_BASES=$bb ASAN_OPTIONS=detect_leaks=0 TSAN_OPTIONS="halt_on_error=0" FS_ORDERLY_EXIT=1 build/asan/fastore_pack_$kind e -itests/golden/$name.in -o$out $flags $gg > $out.log 2>&1
      st=$?
      if [ $st -ne 0 ] || grep -q "ERROR: \|WARNING: ThreadSanitizer\|runtime error" $out.log || ! cmp -s $out.cdata tests/golden/$name.ref.cdata; then echo "FAIL $kind $name $g (exit $st)"; grep -m3 "ERROR\|WARNING\|runtime error" $out.log; rc=1; else echo "ok   $kind $name $g"; fi
      rm -f $out.cdata $out.cmeta
    done
  done < <(python3 -c "
import sys; sys.path.insert(0,'tests')
from conftest import manifest
for n,p,f in manifest(): print(n, int(p), ' '.join(f))")
done
exit $rc
