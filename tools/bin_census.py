#!/usr/bin/env python3
"""bin_census.py <binned prefix> [min_bin_size] -- record counts of the standard bins of a binned library
(what decides the length of the per-bin quality streams, i.e. the tail of a device step)."""
import ctypes as C
import os
import struct
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import fastore_amd

lib = fastore_amd.load_library(os.environ.get("FASTORE_AMD_LIB"))
with fastore_amd.Library(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 256, lib=lib) as L:
    b = L.batch
    raw = C.string_at(b.bins, b.n_bins * 40)
    recs = np.array([struct.unpack_from("<I", raw, 40 * i + 28)[0] for i in range(b.n_bins)], dtype=np.int64)
    recs.sort()
    tot = recs.sum()
    print("std bins %d, records %d, mean %.0f, median %d, max %d" % (len(recs), tot, recs.mean(), np.median(recs), recs.max()))
    cum = np.cumsum(recs[::-1])
    for thr in (2000, 5000, 8000, 10000, 15000, 20000, 25000, 30000, 40000):
        sel = recs[recs > thr]
        print("  bins > %6d reads: %4d bins, %5.1f %% of the records" % (thr, len(sel), 100.0 * sel.sum() / tot))
    print("  largest 16:", recs[::-1][:16].tolist())
    np.save("/tmp/census.npy", recs)
