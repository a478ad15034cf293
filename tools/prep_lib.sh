#!/bin/bash
# prep_lib.sh <workdir> <name> <reads> <len> <seed> [paired]  -- FASTQ -> bin -> rebin x3 with the reference tools (C1 profile)
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$1; N=$2; R=$3; L=$4; S=$5; P=${6:-0}
G=$ROOT/oracle/_ref/ref_driver_gcc; GEN=$ROOT/build/gen_fastq
[ -x "$GEN" ] || g++ -O2 -o "$GEN" $ROOT/tools/gen_fastq.cpp
mkdir -p $W
pe=""; in="$W/${N}_1.fastq"
if [ "$P" = 1 ]; then $GEN --reads $R --len $L --genome $((R*L/50)) --seed $S --paired --out $W/$N; pe="-z"; in="$W/${N}_1.fastq $W/${N}_2.fastq"
else $GEN --reads $R --len $L --genome $((R*L/50)) --seed $S --out $W/$N; fi
$G bin "-i$in" -o$W/$N.b0 -t8 -H -q0 -p8 -s0 -b256 $pe
$G rebin -i$W/$N.b0 -o$W/$N.b2 -t8 -r -w1024 -W1024 -p2 $pe
$G rebin -i$W/$N.b2 -o$W/$N.b4 -t8 -r -w1024 -W1024 -p4 $pe
$G rebin -i$W/$N.b4 -o$W/$N.b8 -t8 -r -w1024 -W1024 -p8 $pe
