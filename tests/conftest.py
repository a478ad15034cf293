import ctypes
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
VECTORS = os.path.join(GOLDEN, "vectors")
REF_DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
REF_DRIVER_GCC = os.path.join(ROOT, "oracle", "_ref", "ref_driver_gcc")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


# The driver gives `pytest -m gpu` 900 s on a fresh box.  The suite is built to end well inside that: every GPU test has a
# timeout of its own (120 s unless it names another), the kernel-level parity tests run first and the tests that make a library
# with the live reference's tools run last, and a session clock fails the NEXT test by name once the run has used its budget --
# a loud failure with a test name in it instead of a silent kill from outside.
GPU_TEST_TIMEOUT_S = 120
GPU_SESSION_BUDGET_S = float(os.environ.get("FS_GPU_SESSION_BUDGET_S", "600"))
_SESSION_T0 = time.monotonic()
_LATE = ("live_reference", "fresh_librar", "long_streams", "many_batches")      # tests that run the reference's tools on the box


def pytest_collection_modifyitems(config, items):
    for it in items:
        if it.get_closest_marker("gpu") and not it.get_closest_marker("timeout"):
            it.add_marker(pytest.mark.timeout(GPU_TEST_TIMEOUT_S, method="thread"))
    # stable: the order inside each class stays the order of the files
    items.sort(key=lambda it: 1 if (it.get_closest_marker("gpu") and any(k in it.name for k in _LATE)) else 0)


def pytest_runtest_setup(item):
    if item.get_closest_marker("gpu"):
        used = time.monotonic() - _SESSION_T0
        if used > GPU_SESSION_BUDGET_S:
            pytest.fail("GPU session clock: %.0f s used of a budget of %.0f s before %s -- the suite must finish inside the driver's window"
                        % (used, GPU_SESSION_BUDGET_S, item.nodeid), pytrace=False)


def _make(target_dir, *args):
    subprocess.check_call(["make", "-C", target_dir, "-j", "8"] + list(args), stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    """The plain-C restatement (oracle/liboracle.so) -- checker only."""
    _make(os.path.join(ROOT, "oracle"), "oracle")
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    lib.fso_ppmd_encode.restype = ctypes.c_size_t
    lib.fso_ppmd_encode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.fso_rc_encode.restype = ctypes.c_size_t
    lib.fso_rc_encode.argtypes = [ctypes.c_int] * 3 + [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    lib.fso_rle_binary.restype = ctypes.c_size_t
    lib.fso_rle_binary.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    lib.fso_rle0.restype = ctypes.c_size_t
    lib.fso_rle0.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    return lib


MODELS = {"s2o4": (0, 1, 4, 0), "s8o4": (1, 3, 4, 0), "a8o4": (2, 3, 4, 1), "a2o10": (3, 1, 10, 1), "a8o6": (4, 3, 6, 1), "a256o1": (5, 8, 1, 1)}


def oracle_ppmd(lib, data):
    buf = ctypes.create_string_buffer(2 * len(data) + 4096)
    n = lib.fso_ppmd_encode(data, len(data), buf, len(buf), None)
    return buf.raw[:n]


def oracle_rc(lib, model, pairs):
    _, bits, order, adv = MODELS[model]
    sym = pairs[0::2]; ctx = pairs[1::2]
    buf = ctypes.create_string_buffer(2 * len(sym) + 64)
    n = lib.fso_rc_encode(bits, order, adv, sym, ctx, len(sym), buf, len(buf))
    return buf.raw[:n]


def oracle_qvz(lib, footer, lens, quals):
    """One block's QVZ quality stream by the oracle (oracle/src/qvz_oracle.c); lens uint32[n], quals uint8[]."""
    import numpy as np
    lens = np.ascontiguousarray(lens, dtype=np.uint32); quals = np.ascontiguousarray(quals, dtype=np.uint8)
    buf = ctypes.create_string_buffer(3 * len(quals) + 64)
    lib.fso_qvz_encode.restype = ctypes.c_long
    lib.fso_qvz_encode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t]
    n = lib.fso_qvz_encode(footer, len(footer), quals.ctypes.data, lens.ctypes.data, len(lens), buf, len(buf))
    assert n >= 0, "oracle rejected the QVZ input"
    return buf.raw[:n]


def manifest():
    out = []
    for line in open(os.path.join(GOLDEN, "manifest.txt")):
        parts = line.split()
        out.append((parts[0], parts[1] == "1", parts[2:]))
    return out


def knobs_from_flags(flags):
    """reference fastore_pack flag letters -> fastore_amd.Packer keyword arguments"""
    m = {"f": "min_bin_size", "c": "min_consensus_size", "d": "max_hamming_distance", "w": "max_lz_window", "W": "max_pair_lz_window",
         "e": "encode_threshold", "E": "pair_encode_threshold", "s": "shift_cost", "m": "mismatch_cost", "q": "max_record_shift_diff",
         "n": "max_new_variants_per_read"}
    kw = dict(extra_reduce_hard_reads=0, extra_reduce_expensive_lz=0)
    for f in flags:
        if f == "-r": kw["extra_reduce_hard_reads"] = 1
        elif f == "-l": kw["extra_reduce_expensive_lz"] = 1
        elif f[1] in m: kw[m[f[1]]] = int(f[2:])
    return kw


def flag_variants():
    """(name, paired, sha256 of the reference .cdata, flags): tests/golden/flag_variants.txt (made by make_golden.sh)"""
    out = []
    for line in open(os.path.join(GOLDEN, "flag_variants.txt")):
        f = line.split()
        if f:
            out.append((f[0], f[1] == "1", f[2], f[3:]))
    return out


C1_FLAGS = ["-r", "-f256", "-c10", "-d8", "-w1024", "-W1024"]          # scripts/fastore_compress.sh:146-148 (the C1 profile's pack flags)
REF_STAGE_TIMEOUT_S = 60


def _ref_stage(argv, threads):
    """One stage of the live reference under a SHORT timeout.  Its multi-threaded stages are known to dead-lock now and then
    (its pack at -t64 and, once in a suite run of round 3, at -t16), so a stage that does not finish in a minute is run
    once more with one worker; a second miss fails the test.  No stage here may wait longer than the test's own timeout."""
    for t in ([threads, 1] if threads > 1 else [1]):
        try:
            subprocess.run([a.replace("@T", str(t)) for a in argv], check=True, timeout=REF_STAGE_TIMEOUT_S, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            return t
        except subprocess.TimeoutExpired:
            continue
    raise RuntimeError("a stage of the reference did not finish in %d s, not even with one worker: %s" % (REF_STAGE_TIMEOUT_S, " ".join(argv[:2])))


def ref_pipeline(tmp, name, reads, length, genome, seed, paired, q, threads=4, gen_flags=()):
    """synthetic FASTQ -> fastore_bin -> 3 x fastore_rebin with the REAL reference (C1 profile); returns (binned prefix, pe flags)"""
    gen = os.path.join(ROOT, "build", "gen_fastq")
    if not os.path.exists(gen):
        os.makedirs(os.path.dirname(gen), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-o", gen, os.path.join(ROOT, "tools", "gen_fastq.cpp")])
    base = os.path.join(tmp, name)
    subprocess.check_call([gen, "--reads", str(reads), "--len", str(length), "--genome", str(genome), "--seed", str(seed), "--out", base] + (["--paired"] if paired else []) + list(gen_flags))
    pe = ["-z"] if paired else []
    inp = base + "_1.fastq" + ((" " + base + "_2.fastq") if paired else "")
    _ref_stage([REF_DRIVER_GCC, "bin", "-i" + inp, "-o" + base + ".b0", "-t@T", "-H", "-q%d" % q, "-p8", "-s0", "-b256"] + pe, threads)
    prev = base + ".b0"
    for p in (2, 4, 8):
        cur = base + ".b%d" % p
        _ref_stage([REF_DRIVER_GCC, "rebin", "-i" + prev, "-o" + cur, "-t@T", "-r", "-w1024", "-W1024", "-p%d" % p] + pe, threads)
        for e in (".bmeta", ".bdna", ".bqua", ".bhead"):          # the stage before is not read again
            if os.path.exists(prev + e):
                os.remove(prev + e)
        prev = cur
    return prev, pe


def reference_pack(binned, out, pe, flags=C1_FLAGS, threads=1):
    """the live reference's pack: -t1 is the archive the product must equal byte for byte; with more workers the blocks
    come in completion order and are compared by signature (reference_blocks)"""
    return _ref_stage([REF_DRIVER, "pack", "-i" + binned, "-o" + out, "-t@T"] + list(flags) + list(pe), threads)


class RefLibs:
    """Libraries made by the live reference's tools, ONE copy per session: several tests read the same binned library (and the same
    reference archive of it), and each costs 5-40 s to make.  Shapes (see `shapes`): a SMALL genome at HIGH coverage gives the bins of
    a BASELINE-sized library -- tens of thousands of reads, quality streams of several million PPMd symbols -- from a few hundred
    thousand reads, which the reference's tools bin in seconds (the standard bins of a library are its signatures, and they grow
    with coverage, not with the genome)."""
    shapes = {
        # name: (reads or pairs, genome, seed, paired, quality mode, generator flags)
        "se_long": (400_000, 6_000, 8, False, 0, ()),            # 39 standard bins, the largest > 30 000 reads = 4.5 M quality symbols
        "pe_long": (200_000, 6_000, 8, True, 0, ()),             # bins of > 10 000 pairs: both mates in one stream, > 3 M symbols
        "pe_noisy": (70_000, 1_500, 11, True, 0, ("--noisy-quality",)),   # one bin of 17 000 pairs of structureless scores (5.1 M symbols): 3 model restarts
    }

    def __init__(self, root):
        self.root = root; self.libs = {}; self.packs = {}

    def library(self, name, shape=None):
        if name not in self.libs:
            reads, genome, seed, paired, q, gen_flags = shape or self.shapes[name]
            d = os.path.join(self.root, name); os.makedirs(d, exist_ok=True)
            binned, pe = ref_pipeline(d, "lib", reads, 150, genome, seed, paired, q, gen_flags=gen_flags)
            self.libs[name] = (binned, pe, d)
        return self.libs[name]

    def packed(self, name, threads=1):
        """(prefix of the reference's archive of the library, workers it was written with)"""
        if name not in self.packs:
            binned, pe, d = self.library(name)
            out = os.path.join(d, "ref")
            self.packs[name] = (out, reference_pack(binned, out, pe, threads=threads))
        return self.packs[name]


@pytest.fixture(scope="session")
def ref_libs(tmp_path_factory):
    if not (os.path.exists(REF_DRIVER) and os.path.exists(REF_DRIVER_GCC)):
        pytest.skip("reference binaries (oracle/_ref) not shipped")
    return RefLibs(str(tmp_path_factory.mktemp("reflibs")))


def reference_blocks(prefix):
    """signature -> block bytes of a reference archive (<prefix>.cmeta footer: u32 n, u64 sizes[n], u32 signatures[n])"""
    import struct
    m = open(prefix + ".cmeta", "rb").read(); d = open(prefix + ".cdata", "rb").read()
    foff, _ = struct.unpack_from("<QQ", m, 0)
    n, = struct.unpack_from("<I", m, foff)
    sizes = struct.unpack_from("<%dQ" % n, m, foff + 4)
    sigs = struct.unpack_from("<%dI" % n, m, foff + 4 + 8 * n)
    out, off = {}, 0
    for sz, sg in zip(sizes, sigs):
        out[sg] = d[off:off + sz]; off += sz
    return out


def check_compress_bins_seam(fastore_amd, packer, name, flags, lib=None):
    """fsgpu_compress_bins on the unpacked standard bins of a golden library == the reference's blocks, bin by bin"""
    kn = knobs_from_flags(flags)
    with fastore_amd.Library(os.path.join(GOLDEN, name + ".in"), kn["min_bin_size"], lib=lib) as L:
        packer.set_archive_params(L.config, L.header_fields, L.quality_codebook)
        blocks = packer.compress_bins(L.batch)
        sigs = L.signatures()
    want = reference_blocks(os.path.join(GOLDEN, name + ".ref"))
    assert len(blocks) == len(sigs) > 20
    assert sorted(sigs) == sigs
    for sg, blk in zip(sigs, blocks):
        assert blk == want[sg], "block of signature %d differs" % sg
    assert len(want) == len(sigs) + 1          # + block 0 (small bins and the N bin), which is not part of this seam


def quality_gather_case(seed, n_strings=400, max_len=310):
    """Random packed quality scores and emitted strings for fs_gather_quality, with the expected stream.
    Restates, in numpy, what the reference leaves in the PPMd input of a lossless bin: FastqPacker stores a score as six
    bits, MSB first (fastore_bin/FastqPacker.cpp:157-287, read back :290-411), and IQualityStoreBase::CompressReadQuality
    (MET_NONE, fastore_pack/FastqCompressor.cpp:229-247) emits score - offset -- the stored six bits -- back to front when the
    record is flagged reverse-complemented."""
    import numpy as np
    rng = np.random.default_rng(seed)
    lens = [0, 1, 2, 3, 4, 5, 63, 64, 65, 150, 151, 255, 256, 257, max_len] + [int(x) for x in rng.integers(0, max_len + 1, n_strings)]
    total = sum(lens) + 3 * len(lens) + 16
    vals = rng.integers(0, 64, total, dtype=np.uint8)
    bits = np.unpackbits(vals[:, None], axis=1)[:, 2:]            # six low bits of every score, MSB first
    packed = np.packbits(bits.reshape(-1)).tobytes()
    strings, expect, pos = [], bytearray(), 0
    for n in lens:
        rev = bool(rng.integers(0, 2))
        strings.append((6 * pos, n, rev))
        seg = vals[pos:pos + n]
        expect += bytes(seg[::-1] if rev else seg)
        pos += n + int(rng.integers(0, 3))                        # strings need not be adjacent in the packed scores
    order = rng.permutation(len(strings))                         # nor emitted in stored order
    out = bytearray()
    chunks = []
    p2 = 0
    for (b, n, r) in strings:
        chunks.append(bytes(expect[p2:p2 + n])); p2 += n
    return packed, [strings[i] for i in order], b"".join(chunks[i] for i in order)


def quality_gather_binned_case(seed, bits, threshold=20, n_strings=300, max_len=255):
    """Random packed 8-bin (3-bit) / binary (1-bit) scores, emitted strings with 'N' positions, and the expected (symbol, context)
    pairs -- numpy restatement of IQualityStoreBase::CompressReadQuality for MET_8BIN / MET_BINARY (fastore_pack/
    FastqCompressor.cpp:249-316): positions under an 'N' are skipped, context = emitted index * 8 (2) / length; a stored bit
    stands for the scores 6 / 40 (FastqPacker) and is coded as score >= threshold; the stored 3-bit value is the bin index."""
    import numpy as np
    rng = np.random.default_rng(seed)
    lens = [1, 2, 3, 7, 8, 9, 63, 64, 65, 150, 151, max_len] + [int(x) for x in rng.integers(1, max_len + 1, n_strings)]
    total = sum(lens) + 3 * len(lens) + 16
    vals = rng.integers(0, 1 << bits, total, dtype=np.uint8)
    b = np.unpackbits(vals[:, None], axis=1)[:, 8 - bits:]
    packed = np.packbits(b.reshape(-1)).tobytes() + b"\0" * 4
    strings, chunks, pos = [], [], 0
    for n in lens:
        rev = bool(rng.integers(0, 2))
        k = int(rng.integers(0, 4)) if n > 4 else 0
        npos = sorted(set(int(x) for x in rng.integers(0, n, k)))
        if len(npos) >= n:
            npos = []
        strings.append((bits * pos, n, rev, bytes(npos)))
        out = bytearray()
        for i in range(n):
            ii = n - 1 - i if rev else i
            if ii in npos:
                continue
            v = int(vals[pos + ii])
            sym = v if bits == 3 else (1 if (40 if v else 6) >= threshold else 0)
            out += bytes([sym, i * (8 if bits == 3 else 2) // n])
        chunks.append(bytes(out))
        pos += n + int(rng.integers(0, 3))
    order = rng.permutation(len(strings))
    return packed, [strings[i] for i in order], b"".join(chunks[i] for i in order)
