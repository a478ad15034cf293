#!/usr/bin/env python3
"""Device QVZ coder micro-benchmark: one long quality block (reads of 60 scores under the test codebook of tests/golden/qvz_inputs.py)
alone and beside copies of itself; kernel time per symbol.  FS_LIB=<alternative build of the library> for A/B runs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import fastore_amd
import qvz_inputs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7_000_000
rng = np.random.default_rng(1)
reads = n // 60
lens = np.full(reads, 60, dtype=np.uint32)
walk = np.clip(38 + np.cumsum(rng.integers(-1, 2, reads * 60)) % 30, 2, 41).astype(np.uint8)
footer = qvz_inputs.qvz_footer()
lib = fastore_amd.load_library(os.environ["FS_LIB"]) if os.environ.get("FS_LIB") else None
with fastore_amd.Packer(lib=lib, device_id=0) as p:
    p.qvz_encode(footer, [(lens[:10], walk[:600])])
    for copies in (1, 64):
        for rep in range(2):
            p.reset_stats(); t = time.perf_counter(); out = p.qvz_encode(footer, [(lens, walk)] * copies); dt = time.perf_counter() - t
            st = p.stats()
            print("copies %3d  %d symbols -> %d bytes  kernel %.1f ms  %.3f us per symbol of one stream  wall %.2f s" % (copies, reads * 60, len(out[0]), st["encode_kernel_ms"], st["encode_kernel_ms"] * 1e3 / (reads * 60), dt), flush=True)
