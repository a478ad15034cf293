#!/usr/bin/env python3
"""Device range-coder micro-benchmark: one long 8-bin quality stream (<8,6>: 150 positions per read, ctx0 = position * 8 / 150)
alone and beside copies of itself.  FS_LIB=<alternative build of the library> for A/B runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastore_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7_000_000
rng = np.random.default_rng(1)
sym = (np.clip(38 + np.cumsum(rng.integers(-1, 2, n)) % 39, 2, 40) // 5 % 8).astype(np.uint8)
ctx = ((np.arange(n) % 150) * 8 // 150).astype(np.uint8)
pairs = np.stack([sym, ctx], 1).tobytes()
lib = fastore_amd.load_library(os.environ["FS_LIB"]) if os.environ.get("FS_LIB") else None
with fastore_amd.Packer(lib=lib, device_id=0) as p:
    p.rc_encode([4], [pairs[:2000]])
    for copies in (1, 64):
        for rep in range(2):
            p.reset_stats(); t = time.perf_counter(); out = p.rc_encode([4] * copies, [pairs] * copies); dt = time.perf_counter() - t
            st = p.stats()
            print("copies %3d  %d symbols -> %d bytes  kernel %.1f ms  %.3f us per symbol of one stream  wall %.2f s" % (copies, n, len(out[0]), st["encode_kernel_ms"], st["encode_kernel_ms"] * 1e3 / n, dt), flush=True)
