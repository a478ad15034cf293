export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2ab
export FS_LIB=$PWD/build/libfastore_amd_ser.so
python3 - <<'PY' > gpurun_out/r2ab_serial_profile.txt 2>&1
import sys, os, time
sys.path.insert(0, '.')
import numpy as np, fastore_amd
rng = np.random.default_rng(1)
def quality(n):
    steps = np.array([-3, -1, 0, 0, 0, 0, 1, 1])[rng.integers(0, 8, n)]
    out = bytearray(n); cur = 38
    for i in range(n):
        if i % 150 == 0: cur = 38
        cur = min(40, max(2, cur + steps[i])); out[i] = cur
    return bytes(out)
base = quality(3000000)
lib = fastore_amd.load_library(os.environ['FS_LIB'])
with fastore_amd.Packer(lib=lib, device_id=0) as p:
    p.ppmd_encode([base[:1000]])
    p.reset_stats(); out = p.ppmd_encode([base]); st = p.stats()
    n = len(base)
    raw = [st[k] for k in ("ppmd_window_attempts", "ppmd_windows", "ppmd_window_symbols", "ppmd_window_rounds", "ppmd_windows_redone")]
    pr = p.window_profile()
    tot = pr["streams_total"]
    print("kernel %.1f ms, %.0f clocks per symbol" % (st["encode_kernel_ms"], 64.0 * tot / n))
    ser = raw[3]
    print("serial symbols %d (%.2f %% of the stream)" % (ser, 100.0 * ser / n))
    print("in rounds: %d swaps, %d positions walked, %d rescales; serial symbols %d; windows (normal build) ~70651" % (raw[0], raw[1], raw[2], raw[3]))
    for k in ("serial_escapes", "serial_update_model", "windows_total"):
        print("  %-40s %6.1f clocks per stream symbol, %7.0f per serial symbol" % (k, 64.0 * pr[k] / n, 64.0 * pr[k] / max(1, ser)))
PY
cat gpurun_out/r2ab_serial_profile.txt
