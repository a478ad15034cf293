#!/usr/bin/env python3
"""How much slower does ONE long PPMd stream run beside many short ones?  (the tail of a device step)
   tools/ppmd_contention.py [long symbols] [short symbols]      env COPIES=0,256,1024,3000 short streams beside the long one"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastore_amd
rng = np.random.default_rng(1)
def quality(n):
    steps = np.array([-3, -1, 0, 0, 0, 0, 1, 1])[rng.integers(0, 8, n)]
    out = bytearray(n); cur = 38
    for i in range(n):
        if i % 150 == 0: cur = 38
        cur = min(40, max(2, cur + steps[i])); out[i] = cur
    return bytes(out)
longs = quality(int(sys.argv[1]) if len(sys.argv) > 1 else 3000000)
shorts = quality(int(sys.argv[2]) if len(sys.argv) > 2 else 600000)
with fastore_amd.Packer(device_id=0, max_waves=int(os.environ.get("MAXW", "0"))) as p:
    p.ppmd_encode([longs[:1000]])
    for copies in [int(c) for c in os.environ.get("COPIES", "0,256,1024,3000").split(",")]:
        p.reset_stats(); out = p.ppmd_encode([longs] + [shorts] * copies)
        st = p.stats()
        print("1 x %d symbols beside %4d x %d: kernel %.1f ms (long stream alone needs ~%.0f ms at 0.15 us/symbol)" % (len(longs), copies, len(shorts), st["encode_kernel_ms"], len(longs) * 0.15e-3), flush=True)
