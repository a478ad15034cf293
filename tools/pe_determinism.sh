#!/bin/bash
# Is the reference's -t32 pack of a library the same archive every time?  Is ours?   tools/pe_determinism.sh <tag> [lib prefix] [runs]
set -u
tag=$1; lib=${2:-/tmp/fastore_bench/pe25000k.b8}; runs=${3:-3}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/${tag}_determinism.txt; : > $out
if [ ! -f $lib.bmeta ]; then echo "no library at $lib" | tee -a $out; exit 3; fi
W=$(dirname $lib)
python3 - >> $out 2>&1 <<PY
import sys, os, time, subprocess, hashlib
sys.path.insert(0, os.getcwd())
import bench, fastore_amd
lib = "$lib"; W = "$W"
def blocks(prefix):
    sizes, sigs = bench.read_archive(prefix)
    d = {}; pos = 0
    with open(prefix + ".cdata", "rb") as f:
        for s, g in zip(sizes, sigs):
            d[int(g)] = hashlib.md5(f.read(s)).hexdigest()
    return d
refs = []
for i in range($runs):
    p = os.path.join(W, "det_ref")
    t = time.time(); subprocess.check_call([bench.REF, "pack", "-i" + lib, "-o" + p, "-t32"] + bench.PACK_FLAGS + ["-z"]); dt = time.time() - t
    refs.append(blocks(p)); print("reference run %d: %.1f s, %d blocks, %d differ from run 0" % (i, dt, len(refs[-1]), sum(1 for g in refs[0] if refs[0][g] != refs[-1].get(g))), flush=True)
ours = []
with fastore_amd.Packer(device_id=0) as pk:
    for i in range($runs):
        p = os.path.join(W, "det_ours")
        t = time.time(); pk.pack_file(lib, p); dt = time.time() - t
        ours.append(blocks(p)); print("our pack %d: %.1f s, %d differ from our pack 0, %d differ from reference run 0" % (i, dt, sum(1 for g in ours[0] if ours[0][g] != ours[-1].get(g)), sum(1 for g in refs[0] if refs[0][g] != ours[-1].get(g))), flush=True)
PY
cat $out
