#!/bin/bash
# Which blocks of a packed library differ from the reference's, and under which switches they do not.   tools/pe_parity_debug.sh <tag> [lib prefix]
set -u
tag=$1; lib=${2:-/tmp/fastore_bench/pe25000k.b8}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/${tag}_parity_debug.txt; : > $out
if [ ! -f $lib.bmeta ]; then echo "no library at $lib" | tee -a $out; exit 3; fi
W=$(dirname $lib)
python3 - >> $out 2>&1 <<PY
import sys, os, time, subprocess, struct
sys.path.insert(0, os.getcwd())
import bench, fastore_amd
lib = "$lib"; W = "$W"
refp = os.path.join(W, "dbg_ref")
if not os.path.exists(refp + ".cmeta"):
    t = time.time(); subprocess.check_call([bench.REF, "pack", "-i" + lib, "-o" + refp, "-t32"] + bench.PACK_FLAGS + ["-z"]); print("reference packed in %.1f s" % (time.time() - t), flush=True)
def diff(ours):
    so, go = bench.read_archive(ours); sr, gr = bench.read_archive(refp)
    off, pos = {}, 0
    for s, g in zip(sr, gr): off[g] = (pos, s); pos += s
    bad = []
    with open(ours + ".cdata", "rb") as fo, open(refp + ".cdata", "rb") as fr:
        for s, g in zip(so, go):
            p, rs = off[g]; fr.seek(p); a = fo.read(s); b = fr.read(rs)
            if a != b:
                k = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
                bad.append((g, s, rs, k))
    return bad
variants = ({}, {"FS_SEARCH_SURPLUS": "0"}, {"FS_DEVICE_MATCHER": "0"}, {"FS_DEVICE_EMIT": "0"}, {"FS_SPLIT_PIPELINES": "0"}, {"FS_WG_BUDGET": "0"})
for env in variants[:int(os.environ.get("PARITY_VARIANTS", "6"))]:
    for k in ("FS_SEARCH_SURPLUS", "FS_DEVICE_MATCHER", "FS_DEVICE_EMIT", "FS_SPLIT_PIPELINES", "FS_WG_BUDGET"): os.environ.pop(k, None)
    os.environ.update(env)
    o = os.path.join(W, "dbg_out")
    with fastore_amd.Packer(device_id=0) as p:
        t = time.time(); p.pack_file(lib, o); dt = time.time() - t
    bad = diff(o)
    print("env %s: %.1f s, %d blocks differ: %s" % (env, dt, len(bad), bad[:6]), flush=True)
PY
cat $out
