#!/usr/bin/env python3
"""Device PPMd micro-benchmark: per-symbol latency of ONE stream and throughput of many copies."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastore_amd
rng = np.random.default_rng(1)
def quality(n):
    steps = np.array([-3, -1, 0, 0, 0, 0, 1, 1])[rng.integers(0, 8, n)]
    q = np.empty(n, dtype=np.int64); cur = 38
    out = bytearray(n); L = 150
    for i in range(n):
        if i % L == 0: cur = 38
        cur = min(40, max(2, cur + steps[i])); out[i] = cur
    return bytes(out)
base = quality(int(sys.argv[1]) if len(sys.argv) > 1 else 300000)
_lib = fastore_amd.load_library(os.environ['FS_LIB']) if os.environ.get('FS_LIB') else None
with fastore_amd.Packer(lib=_lib, device_id=0, max_waves=int(sys.argv[2]) if len(sys.argv) > 2 else 0) as p:
    p.ppmd_encode([base[:1000]])
    for copies in [int(c) for c in os.environ.get("COPIES", "1,600,3000,6144,12288").split(",")]:
        p.reset_stats(); t = time.perf_counter(); out = p.ppmd_encode([base] * copies); dt = time.perf_counter() - t
        st = p.stats()
        print("copies %5d  len %d -> %d  kernel %.1f ms  per-symbol(one stream) %.2f us  aggregate %.1f Msym/s  wall %.2f s" % (
            copies, len(base), len(out[0]), st["encode_kernel_ms"], st["encode_kernel_ms"] * 1e3 / len(base), copies * len(base) / st["encode_kernel_ms"] / 1e3, dt), flush=True)
        w = max(1, st["ppmd_windows"])
        print("   windows: %.1f %% of the symbols in %d windows (%.1f symbols, %.2f rounds per window, %.2f of the windows without a round; %d attempts, %d redone)" % (
            100.0 * st["ppmd_window_symbols"] / max(1, st["ppmd_symbols"]), st["ppmd_windows"], st["ppmd_window_symbols"] / w, st["ppmd_window_rounds"] / w, st["ppmd_window_light_rounds"] / w,
            st["ppmd_window_attempts"], st["ppmd_windows_redone"]), flush=True)
        print("   %d rescales inside windows let states drop out; coder tail launches %d" % (st.get("ppmd_window_drops", 0), st.get("coder_tail_launches", 0)), flush=True)
        if "serprof" in os.environ.get("FS_LIB", ""):      # -DFS_SER_PROFILE builds: the serial path's clocks ride in the window counters' places
            ns = max(1, st["ppmd_window_rounds"])
            print("   serial path (FS_SER_PROFILE): %d serial symbols; clocks per serial symbol: start -> first context ready %.0f, first-context coding + hand-off %.0f, tail of the loop %.0f; failed window attempts %.0f" % (
                ns, 64.0 * st["ppmd_window_attempts"] / ns, 64.0 * st["ppmd_windows"] / ns, 64.0 * st["ppmd_window_symbols"] / ns, 64.0 * st["ppmd_windows_redone"] / ns), flush=True)
        pr = p.window_profile()
        if pr["streams_total"]:
            tot = pr["streams_total"]
            print("   clocks per symbol: %.0f ; shares: " % (64.0 * tot / (copies * len(base))) + ", ".join("%s %.1f %%" % (k, 100.0 * v / tot) for k, v in pr.items() if k != "streams_total"), flush=True)
