"""Seeded inputs of the QVZ golden vectors (shared by make_vectors.py and the tests, so that the large
rescale case needs only its expected output committed)."""
import os, struct
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def qvz_footer():
    """The quality section (WELL state, max read length, codebook) of the reference-made se_qvz archive footer."""
    m = open(os.path.join(HERE, "se_qvz.ref.cmeta"), "rb").read()
    foff, fsize = struct.unpack_from("<QQ", m, 0)
    n, = struct.unpack_from("<I", m, foff)
    return m[foff + 4 + 12 * n + 56: foff + fsize]        # the oracle / reference parse stops at the section's end


def wide_footer():
    """Synthetic 3-column codebook whose quantizers keep 72 / 36 distinct values (output alphabets beyond 64 symbols):
    the wide-alphabet paths no trained codebook reaches."""
    rng = np.random.default_rng(79)
    ident = np.arange(72, dtype=np.uint8)
    pairs = (ident // 2 * 2).astype(np.uint8)                    # 36 distinct values
    def line(q): return bytes((q.astype(np.uint16) + 33).astype(np.uint8))
    out = bytearray()
    out += rng.integers(0, 2**32, 32, dtype=np.uint64).astype("<u4").tobytes()      # WELL state
    out += struct.pack("<I", 3)                                                      # max_read_length = columns
    out += bytes([64 + 33]) + struct.pack(">H", 72) + line(ident) + line(pairs)      # column 0: ratio, size, low, high
    for _ in (1, 2):                                                                  # input alphabet = all 72 symbols
        out += struct.pack(">H", 72) + bytes((rng.integers(0, 128, 72) + 33).astype(np.uint8))
        for hl in (0, 1):
            for i in range(72):
                out += line(ident if (i + hl) % 3 else pairs)
    return bytes(out)


def footer_for(name):
    return wide_footer() if name == "qvz_wide" else qvz_footer()


def _walk(rng, n):
    steps = rng.integers(0, 8, n); tbl = np.array([-3, -1, 0, 0, 0, 0, 1, 1])
    q = np.empty(n, dtype=np.uint8); cur = 36
    for i, s in enumerate(steps):
        cur = min(41, max(2, cur + tbl[s])); q[i] = cur
    return q


def reads_case(name):
    """-> (lens uint32[n], quality values uint8[sum lens])"""
    if name == "qvz_reads":
        rng = np.random.default_rng(77)
        lens = rng.integers(1, 61, 400).astype(np.uint32)
        return lens, _walk(rng, int(lens.sum()))
    if name == "qvz_rescale":          # > 65536 uses of the two column-0 contexts: exercises the halve-plus-one rescaling
        rng = np.random.default_rng(78)
        lens = rng.integers(1, 3, 150000).astype(np.uint32)
        return lens, rng.integers(2, 42, int(lens.sum())).astype(np.uint8)
    if name == "qvz_wide":
        rng = np.random.default_rng(80)
        lens = rng.integers(1, 4, 30000).astype(np.uint32)
        return lens, rng.integers(0, 72, int(lens.sum())).astype(np.uint8)
    if name == "qvz_one":
        return np.array([1], dtype=np.uint32), np.array([30], dtype=np.uint8)
    raise KeyError(name)


def reads_blob(lens, quals):
    return struct.pack("<I", len(lens)) + lens.astype("<u4").tobytes() + quals.tobytes()

CASES = ("qvz_reads", "qvz_rescale", "qvz_one", "qvz_wide")
