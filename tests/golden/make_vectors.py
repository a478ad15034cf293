#!/usr/bin/env python3
"""Golden vectors for the entropy coders, produced by the REAL reference (oracle/_ref/ref_driver
ppmd|rc, i.e. PpmdEncoder and TEncoder<...> compiled from /root/reference).  Inputs are seeded
numpy streams; only data is stored (vectors/<name>.in, vectors/<name>.out)."""
import os, subprocess, numpy as np
here = os.path.dirname(os.path.abspath(__file__)); R = os.path.join(here, "../../oracle/_ref/ref_driver"); V = os.path.join(here, "vectors")
rng = np.random.default_rng(2024)
def walk(n):  # quality-like bounded random walk
    q = 38; out = bytearray()
    steps = rng.integers(0, 8, n); tbl = [-3, -1, 0, 0, 0, 0, 1, 1]
    for s in steps:
        q = min(40, max(2, q + tbl[s])); out.append(q)
    return bytes(out)
ppmd = {"ppmd_one": b"A", "ppmd_two": b"AB", "ppmd_same1000": b"A" * 1000, "ppmd_qual20k": walk(20000),
        "ppmd_dna8k": bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 8000)), "ppmd_rand4k": rng.integers(0, 256, 4096, dtype=np.uint8).tobytes(),
        "ppmd_flags3k": bytes(rng.choice(np.array([0, 1, 2, 3, 3, 3, 6, 7, 8], dtype=np.uint8), 3000))}
for k, d in ppmd.items():
    open(f"{V}/{k}.in", "wb").write(d)
    subprocess.check_call([R, "ppmd", f"{V}/{k}.in", f"{V}/{k}.out"])
models = {"s2o4": 2, "s8o4": 8, "a8o4": 8, "a2o10": 2, "a8o6": 8, "a256o1": 256}
for m, A in models.items():
    n = 3000
    sym = np.minimum(rng.geometric(0.35, n) - 1, A - 1).astype(np.uint8) if A > 2 else rng.integers(0, 2, n, dtype=np.uint8)
    ctx = rng.integers(0, min(A, 8), n, dtype=np.uint8)
    inter = np.empty(2 * n, dtype=np.uint8); inter[0::2] = sym; inter[1::2] = ctx
    open(f"{V}/rc_{m}.in", "wb").write(inter.tobytes())
    subprocess.check_call([R, "rc", m, f"{V}/rc_{m}.in", f"{V}/rc_{m}.out"])
open(f"{V}/rc_empty.in", "wb").write(b"")
subprocess.check_call([R, "rc", "a256o1", f"{V}/rc_empty.in", f"{V}/rc_empty.out"])
print(sorted(os.listdir(V)))
# ---- QVZ (--lossy) quality streams: the reference's codebook reader, WELL generator, quantizer choice and
# arithmetic coder (`ref_driver qvz`) on seeded reads; footer = the one of the se_qvz golden archive ----
import tempfile
from qvz_inputs import CASES, footer_for, reads_case, reads_blob
with tempfile.TemporaryDirectory() as T:
    for name in CASES:
        open(f"{T}/footer", "wb").write(footer_for(name))
        lens, quals = reads_case(name)
        open(f"{T}/reads", "wb").write(reads_blob(lens, quals))
        subprocess.check_call([R, "qvz", f"{T}/footer", f"{T}/reads", f"{V}/{name}.out"])
print(sorted(os.listdir(V)))
