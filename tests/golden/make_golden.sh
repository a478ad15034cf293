#!/bin/bash
# Regenerates the golden fixtures: small synthetic libraries run through the REAL reference
# (oracle/_ref, built from /root/reference by oracle/Makefile): fastore_bin -> 3 x fastore_rebin
# (C1 profile of scripts/fastore_compress.sh:146-148,186-209) give the binned INPUT of the hot path,
# `pack -t1` (clang build, canonical PE semantics) gives the EXPECTED archive.
# Only data is stored: <name>.in.{bmeta,bdna,bqua,bhead} and <name>.ref.{cdata,cmeta}.
set -euo pipefail
cd "$(dirname "$0")"
ROOT=../..
G=$ROOT/oracle/_ref/ref_driver_gcc
R=$ROOT/oracle/_ref/ref_driver
GEN=$ROOT/build/gen_fastq
[ -x "$GEN" ] || g++ -O2 -o "$GEN" $ROOT/tools/gen_fastq.cpp
T=$(mktemp -d)
# -f24: bins of >= 24 records are "standard" (LZ) bins, so that these small libraries exercise the LZ path and block 0
PACKFLAGS="-r -f24 -c10 -d8 -w1024 -W1024"
ONLY=${1:-}                     # optional: regenerate just this fixture
[ -z "$ONLY" ] && rm -f manifest.txt
make_one() {   # name reads len genome seed paired qflag headerflags
    local name=$1 reads=$2 len=$3 genome=$4 seed=$5 paired=$6 q=$7 hf=$8
    if [ -n "$ONLY" ] && [ "$ONLY" != "$name" ]; then return; fi
    local pe="" in="$T/$name"_1.fastq
    if [ "$paired" = 1 ]; then $GEN --reads $reads --len $len --genome $genome --seed $seed --paired --out $T/$name; pe="-z"; in="$T/${name}_1.fastq $T/${name}_2.fastq"
    else $GEN --reads $reads --len $len --genome $genome --seed $seed --out $T/$name; fi
    $G bin "-i$in" -o$T/$name.b0 -t2 $hf -q$q -p8 -s0 -b256 $pe
    $G rebin -i$T/$name.b0 -o$T/$name.b2 -t2 -r -w1024 -W1024 -p2 $pe
    $G rebin -i$T/$name.b2 -o$T/$name.b4 -t2 -r -w1024 -W1024 -p4 $pe
    $G rebin -i$T/$name.b4 -o$T/$name.b8 -t2 -r -w1024 -W1024 -p8 $pe
    $R pack -i$T/$name.b8 -o$T/$name.ref -t1 $PACKFLAGS $pe
    grep -q "^$name " manifest.txt 2>/dev/null || echo "$name $paired $PACKFLAGS" >> manifest.txt
    for e in bmeta bdna bqua; do cp $T/$name.b8.$e $name.in.$e; done
    [ -f $T/$name.b8.bhead ] && cp $T/$name.b8.bhead $name.in.bhead
    cp $T/$name.ref.cdata $name.ref.cdata; cp $T/$name.ref.cmeta $name.ref.cmeta
    # the reference's -v statistics on stdout (CompressorModule.cpp:357-387), for the two lossless fixtures
    case $name in se_lossless|pe_lossless) $R pack -i$T/$name.b8 -o$T/$name.v -t1 -v $PACKFLAGS $pe > $name.ref.vout 2>/dev/null;; esac
}
#        name          reads len genome seed pe q  headers
make_one se_lossless   9000  100 18000  11   0  0  "-H"
make_one pe_lossless   4000  100 16000  12   1  0  "-H"
make_one se_reduced    5000  100 10000  13   0  2  "-H -C"
make_one se_noheader   4000  80  6400   14   0  0  ""
make_one se_binary     4000  100 8000   15   0  1  ""
make_one se_qvz        3000  60  3600   16   0  3  "-H"       # --lossy: QVZ codebook + WELL seed in the footer
# Non-default matcher / consensus flags on the fixtures above: only the SHA-256 of the reference's .cdata is kept.
# (The reference itself crashes on these inputs with -e below the automatic threshold, -s2 and, SE, -m9: not listed.)
if [ -z "$ONLY" ] || [ "$ONLY" = flag_variants ]; then
    : > flag_variants.txt
    variant() {   # name paired flags...
        local name=$1 paired=$2; shift 2
        local pe=""; [ "$paired" = 1 ] && pe="-z"
        $R pack -i$name.in -o$T/fv -t1 "$@" $pe
        echo "$name $paired $(sha256sum < $T/fv.cdata | cut -d' ' -f1) $*" >> flag_variants.txt
    }
    variant se_lossless 0 -r -l -f24 -c3 -d20 -w1024 -W1024 -q3 -n1
    variant se_lossless 0 -f24 -c10 -d8 -w8 -W4
    variant se_lossless 0 -r -f24 -c2 -d1 -w1024 -W1024 -e80 -E80
    variant pe_lossless 1 -r -l -f24 -c3 -d20 -w1024 -W1024 -q3 -n1 -m9
    variant pe_lossless 1 -f24 -c10 -d8 -w8 -W4
    variant pe_lossless 1 -r -f40 -c2 -d1 -w1024 -W1024 -e80 -E80
    variant se_reduced 0 -r -l -f24 -c2 -d1 -w16 -W16 -q1 -n0 -m9
    variant se_qvz 0 -l -f30 -c4 -d3 -w1024 -W1024
fi
# C0 profile of scripts/fastore_compress.sh: fastore_bin output packed directly (the bin-stage writer flavour of the
# record grammar, no read groups; no -r; 256-entry windows)
if [ -z "$ONLY" ] || [ "$ONLY" = se_c0 ]; then
    C0FLAGS="-f24 -c10 -d8 -w256 -W256"
    $GEN --reads 3000 --len 100 --genome 6000 --seed 23 --out $T/c0
    $G bin -i$T/c0_1.fastq -o$T/c0.b0 -t1 -H -q0 -p8 -s0 -b256
    $R pack -i$T/c0.b0 -o$T/c0.ref -t1 $C0FLAGS
    grep -q "^se_c0 " manifest.txt 2>/dev/null || echo "se_c0 0 $C0FLAGS" >> manifest.txt
    for e in bmeta bdna bqua bhead; do cp $T/c0.b0.$e se_c0.in.$e; done
    cp $T/c0.ref.cdata se_c0.ref.cdata; cp $T/c0.ref.cmeta se_c0.ref.cmeta
fi
# Reads of different lengths: the reference pack (and rebin) index their consensus buffers by the first read's length and
# crash on such a library, so there is no expected archive -- the fixture is the bin-stage output only (bin -t1: -t2 races)
# and the test expects a clean error from the product.
if [ -z "$ONLY" ] || [ "$ONLY" = se_varlen ]; then
    $GEN --reads 2500 --len 100 --genome 5000 --seed 17 --out $T/vl
    awk 'NR%4==1{n=60+(int((NR-1)/4)*37)%41} NR%4==2||NR%4==0{print substr($0,1,n);next}{print}' $T/vl_1.fastq > $T/varlen_1.fastq
    $G bin -i$T/varlen_1.fastq -o$T/varlen.b0 -t1 -q0 -p8 -s0 -b256
    for e in bmeta bdna bqua; do cp $T/varlen.b0.$e se_varlen.in.$e; done
fi
rm -rf "$T"
ls -la
