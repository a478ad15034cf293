"""The 64-lane code paths of the PPMd core (the ones the device runs: lane-parallel scans and the windowed hit path of
ppmd_window.h) on the lock-step wave emulation of tests/emu/simt.h, against the oracle's C restatement.  No GPU needed:
the same source is compiled with -DFS_SIMT_EMU and runs as 64 cooperative fibers."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, VECTORS, manifest, oracle_ppmd


@pytest.fixture(scope="module")
def simt():
    out = os.path.join(ROOT, "build", "libsimt_emu.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    # FS_EMU_DEFS: extra -D flags (kernel experiments, e.g. -DFS_WIN_PREFETCH=1) for the same tests
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DFS_SIMT_EMU"] + os.environ.get("FS_EMU_DEFS", "").split() + ["-shared", "-fPIC", "-o", out,
                           os.path.join(ROOT, "tests", "emu", "ppmd_simt.cpp"), os.path.join(ROOT, "tests", "emu", "mates_simt.cpp"), os.path.join(ROOT, "tests", "emu", "emit_simt.cpp"), os.path.join(ROOT, "tests", "emu", "simt.cpp"), os.path.join(ROOT, "tests", "emu", "qvz_host_ref.cpp")])
    lib = ctypes.CDLL(out)
    lib.simt_ppmd_encode.restype = ctypes.c_size_t
    lib.simt_ppmd_encode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    lib.simt_ppmd_encode_two_waves.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5
    lib.simt_quick_rescales.restype = ctypes.c_ulonglong
    return lib


def encode(lib, data):
    buf = ctypes.create_string_buffer(2 * len(data) + 4096)
    st = (ctypes.c_uint64 * 8)()
    # the device batch pads every stream to 16 bytes: the window reads whole aligned words around the last symbols
    n = lib.simt_ppmd_encode(data + b"\0" * 32, len(data), buf, len(buf), None, st)
    return buf.raw[:n], {"attempts": st[0], "windows": st[1], "covered": st[2], "rounds": st[3], "redone": st[4], "drops": st[7]}


def quality(n, seed, read_len=150):
    rng = np.random.default_rng(seed)
    steps = np.array([-3, -1, 0, 0, 0, 0, 1, 1])[rng.integers(0, 8, n)]
    out = np.empty(n, np.uint8); cur = 38
    for i in range(n):
        if i % read_len == 0:
            cur = 38
        cur = min(40, max(2, cur + steps[i])); out[i] = cur
    return out.tobytes()


def test_windowed_hit_path_reproduces_the_serial_walk_on_quality_streams(simt, oracle):
    data = quality(300_000, 1)
    got, st = encode(simt, data)
    assert got == oracle_ppmd(oracle, data)
    # the path under test really ran: most of the stream went through windows, with shared contexts and redone windows
    assert st["covered"] > 0.7 * len(data) and st["rounds"] > st["windows"] and st["redone"] > 0, st
    # rescales that let states drop out of their context were taken inside windows (their units shrunk at the window's commit)
    assert st["drops"] > 100, st
    # most rescales inside the rounds take the short form (no sorting network); the emulation build holds every one of them
    # against the network and aborts on a difference
    assert simt.simt_quick_rescales() > 1000


@pytest.mark.parametrize("n", [1, 5, 63, 64, 65, 100, 4097, 20_001])
def test_window_edges_short_and_ragged_streams(simt, oracle, n):
    for data in (quality(n, n, read_len=37), bytes([30]) * n, bytes([7, 9]) * (n // 2) + bytes([7]) * (n % 2)):
        got, _ = encode(simt, data)
        assert got == oracle_ppmd(oracle, data)


def test_other_stream_kinds_through_the_64_lane_paths(simt, oracle):
    rng = np.random.default_rng(3)
    streams = {
        "bases": rng.choice(np.frombuffer(b"ACGTN.", dtype=np.uint8), 60_000).tobytes(),
        "flags": rng.integers(0, 9, 40_000).astype(np.uint8).tobytes(),
        "noise": rng.integers(0, 256, 12_000).astype(np.uint8).tobytes(),
        "skewed": np.minimum(rng.geometric(0.4, 80_000) - 1, 255).astype(np.uint8).tobytes(),      # many states per context
        "periodic": (bytes(range(40)) * 2000),                                                    # every context binary
    }
    for name, data in streams.items():
        got, st = encode(simt, data)
        assert got == oracle_ppmd(oracle, data), name


def test_reference_vectors_through_the_64_lane_paths(simt):
    names = sorted(f[:-3] for f in os.listdir(VECTORS) if f.startswith("ppmd_") and f.endswith(".in"))
    assert names
    for nme in names:
        data = open(os.path.join(VECTORS, nme + ".in"), "rb").read()
        if len(data) > 400_000:
            continue
        got, _ = encode(simt, data)
        assert got == open(os.path.join(VECTORS, nme + ".out"), "rb").read(), nme


def encode_two_waves(lib, streams):
    """the two-wave form: a model wave and a coder wave (64 fibers each) with the coding steps queued through LDS; the
    streams go through the same pair of waves one after the other, as the items of a launch do"""
    k = len(streams)
    pads = [s + b"\0" * 32 for s in streams]
    ins = (ctypes.c_char_p * k)(*pads); lens = (ctypes.c_size_t * k)(*[len(s) for s in streams])
    bufs = [ctypes.create_string_buffer(2 * len(s) + 4096) for s in streams]
    outs = (ctypes.c_void_p * k)(*[ctypes.addressof(b) for b in bufs]); caps = (ctypes.c_size_t * k)(*[len(b) for b in bufs])
    sizes = (ctypes.c_uint32 * k)()
    lib.simt_ppmd_encode_two_waves(k, ins, lens, outs, caps, sizes)
    return [bufs[i].raw[:sizes[i]] for i in range(k)]


def test_two_wave_form_model_wave_and_coder_wave(simt, oracle):
    rng = np.random.default_rng(9)
    streams = [quality(120_000, 2), b"", rng.choice(np.frombuffer(b"ACGTN.", dtype=np.uint8), 20_000).tobytes(), b"Q",
               rng.integers(0, 256, 6_000).astype(np.uint8).tobytes(), rng.integers(0, 9, 15_000).astype(np.uint8).tobytes(),
               quality(777, 5), bytes(range(40)) * 300, quality(64, 6), quality(65, 7)]
    got = encode_two_waves(simt, streams)
    for i, (s, g) in enumerate(zip(streams, got)):
        assert g == (oracle_ppmd(oracle, s) if s else b""), i


def rc_encode(lib, model, pairs):
    lib.simt_rc_encode.restype = ctypes.c_size_t
    lib.simt_rc_encode.argtypes = [ctypes.c_uint, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    buf = ctypes.create_string_buffer(len(pairs) + 4096)
    n = lib.simt_rc_encode(model, pairs + b"\0" * 256, len(pairs) // 2, buf, len(buf))
    return None if n == ctypes.c_size_t(-1).value else buf.raw[:n]


@pytest.mark.parametrize("model", ["s2o4", "s8o4", "a8o4", "a2o10", "a8o6"])
def test_windowed_range_coder_reproduces_the_one_symbol_loop(simt, oracle, model):
    # encode_stream_windowed (64 symbols per step: rows fetched side by side, shared rows by rank, write-back by the last
    # position, the range coder in stream order) against the oracle's restatement of the reference's coder stack
    from conftest import MODELS, oracle_rc
    mid, bits, order, adv = MODELS[model]
    rng = np.random.default_rng(5 + mid)
    A = 1 << bits
    cases = []
    for n in (0, 1, 2, 63, 64, 65, 127, 1000, 40_000):
        sym = rng.integers(0, A, n).astype(np.uint8)
        ctx = rng.integers(0, A if adv else 1, n).astype(np.uint8)
        cases.append((sym, ctx))
    # quality-like: long runs of one symbol in few contexts -- the same row again and again inside a window, and often enough
    # to reach the rescale limit (65 536 - 8 A: a row of a two-symbol alphabet is due after 8 190 hits)
    n = 60_000
    walk = np.clip(np.cumsum(rng.integers(-1, 2, n)) // 40 % A, 0, A - 1).astype(np.uint8)
    pos = ((np.arange(n) % 150) * A // 150).astype(np.uint8) if adv else np.zeros(n, np.uint8)
    cases.append((walk, pos))
    cases.append((np.zeros(30_000, np.uint8), np.zeros(30_000, np.uint8)))              # one row only: a rescale every few thousand symbols
    cases.append((np.full(20_000, A - 1, np.uint8), np.full(20_000, (A - 1) if adv else 0, np.uint8)))
    for sym, ctx in cases:
        pairs = np.stack([sym, ctx], axis=1).astype(np.uint8).tobytes()
        assert rc_encode(simt, mid, pairs) == oracle_rc(oracle, model, pairs), (model, len(sym))
    # a symbol outside the alphabet, a context outside its field: the stream is given up as in the one-symbol loop
    bad = bytearray(np.stack([np.zeros(200, np.uint8), np.zeros(200, np.uint8)], axis=1).tobytes()); bad[2 * 130] = A
    assert rc_encode(simt, mid, bytes(bad)) is None


def qvz_case(rng, n_ctx, n, max_card, same_ctx_runs):
    """a synthetic model blob (ModelHeader | Desc[n_ctx] | image: per context its total and its counts) and n symbols ctx | x << 24"""
    cards = rng.integers(1, max_card + 1, n_ctx).astype(np.uint32)
    offs = np.concatenate([[0], np.cumsum(cards + 1)[:-1]]).astype(np.uint32)
    image = []
    for c in cards:
        counts = rng.integers(1, 40, int(c)).astype(np.uint32)
        image += [int(counts.sum())] + [int(v) for v in counts]
    image = np.array(image, np.uint32)
    blob = np.array([n_ctx, len(image), 0, 0], np.uint32).tobytes() + np.stack([offs, cards], 1).astype(np.uint32).tobytes() + image.tobytes()
    ctx = rng.integers(0, n_ctx, n).astype(np.uint32)
    if same_ctx_runs:                                   # the same context several times in a row, and one hot context: shared windows, rescales
        for i in range(0, n - 8, 97):
            ctx[i:i + int(rng.integers(2, 8))] = ctx[i]
        ctx[rng.random(n) < 0.5] = 0
    x = (rng.integers(0, 1 << 30, n) % cards[ctx]).astype(np.uint32)
    return blob, (ctx | (x << 24)).astype(np.uint32).tobytes(), 4 * len(image)


def test_qvz_forms_on_the_emulated_wave_reproduce_the_one_lane_coder(simt):
    # the QVZ coder's 64-lane forms -- a symbol per trip with the lanes over the context's counts (0), 64 symbols per step (1: the one-wave
    # kernels), and that with the fractions and the interval's pass on the coder wave of the two-wave kernels (2: the symbols' counts through
    # the PPMd walk's ring, two emulated wavefronts) -- against the same header compiled for one lane
    for f in (simt.simt_qvz_encode, simt.host_qvz_encode):
        f.restype = ctypes.c_long
    simt.simt_qvz_encode.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    simt.host_qvz_encode.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    rng = np.random.default_rng(77)
    for n_ctx, n, max_card, runs in ((300, 0, 8, False), (300, 1, 8, False), (300, 63, 8, False), (300, 64, 8, False), (300, 65, 8, False), (300, 20_000, 8, False),
                                     (40, 30_000, 72, True), (3, 150_000, 5, True)):
        blob, syms, arena = qvz_case(rng, n_ctx, n, max_card, runs)
        cap = 4 * n + 64
        want = ctypes.create_string_buffer(cap); nw = simt.host_qvz_encode(blob, syms + b"\0" * 256, n, arena, want, cap)
        assert nw >= 0
        for form in (0, 1, 2):
            got = ctypes.create_string_buffer(cap); ng = simt.simt_qvz_encode(form, blob, syms + b"\0" * 256, n, arena, got, cap)
            assert ng == nw and got.raw[:ng] == want.raw[:nw], (form, n_ctx, n)
    # a symbol outside its context's alphabet gives the stream up in every form
    blob, syms, arena = qvz_case(rng, 50, 500, 6, False)
    bad = bytearray(syms); bad[4 * 300 + 3] = 200
    out = ctypes.create_string_buffer(4096)
    assert simt.host_qvz_encode(blob, bytes(bad) + b"\0" * 256, 500, arena, out, 4096) == -1
    for form in (0, 1, 2):
        assert simt.simt_qvz_encode(form, blob, bytes(bad) + b"\0" * 256, 500, arena, out, 4096) == -1


def test_range_coded_streams_through_the_coder_wave(simt, oracle):
    # the two-wave kernel with the windowed range coders (fs_encode_streams2_w): they send their triples through the PPMd walk's
    # ring and the coder wave codes them (coder_wave<true>); PPMd members in between -- the coder wave keeps two range coders
    # apart.  Against the oracle's coders, on two emulated waves.
    from conftest import MODELS, oracle_rc
    lib = simt
    lib.simt_rc_encode_two_waves.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 6
    rng = np.random.default_rng(41)
    streams = []                                        # (model id or None, bytes)
    for name in ("a8o6", "s2o4", "a2o10", "a8o4", "s8o4"):
        mid, bits, order, adv = MODELS[name]
        A = 1 << bits
        for n in (0, 1, 64, 65, 3000):
            sym = rng.integers(0, A, n).astype(np.uint8); ctx = rng.integers(0, A if adv else 1, n).astype(np.uint8)
            streams.append((name, np.stack([sym, ctx], 1).astype(np.uint8).tobytes()))
        n = 40_000                                      # quality-like: shared rows inside windows, rescales
        sym = np.clip(np.cumsum(rng.integers(-1, 2, n)) // 40 % A, 0, A - 1).astype(np.uint8)
        ctx = ((np.arange(n) % 150) * A // 150).astype(np.uint8) if adv else np.zeros(n, np.uint8)
        streams.append((name, np.stack([sym, ctx], 1).tobytes()))
        streams.append((None, quality(3000, 9)))        # a PPMd member between the range-coded streams
    streams.append(("a8o6", np.stack([np.zeros(30_000, np.uint8), np.zeros(30_000, np.uint8)], 1).tobytes()))
    k = len(streams)
    models = (ctypes.c_uint * k)(*[0xFFFFFFFF if m is None else MODELS[m][0] for m, _ in streams])
    ins = [ctypes.create_string_buffer(d + b"\0" * 256) for _, d in streams]
    lens = (ctypes.c_size_t * k)(*[len(d) if m is None else len(d) // 2 for m, d in streams])
    bufs = [ctypes.create_string_buffer(2 * len(d) + 4096) for _, d in streams]
    caps = (ctypes.c_size_t * k)(*[len(b) for b in bufs])
    sizes = (ctypes.c_uint32 * k)()
    inp = (ctypes.c_void_p * k)(*[ctypes.addressof(b) for b in ins]); outp = (ctypes.c_void_p * k)(*[ctypes.addressof(b) for b in bufs])
    assert lib.simt_rc_encode_two_waves(k, models, inp, lens, outp, caps, sizes) == 0
    for i, (m, d) in enumerate(streams):
        want = oracle_ppmd(oracle, d) if m is None else oracle_rc(oracle, m, d)
        assert bufs[i].raw[:sizes[i]] == want, (i, m, len(d))


def test_mate_search_kernel_body_with_a_short_alignment_list(monkeypatch):
    # the same with a list of 256 alignments instead of 4 096 (-DFSM_ITEMS=256): the first pass overflows, the pair before is "dense", the guess
    # is too low -- every pass over a part of the history at a time that a deep library needs (pe_long on the device) is met on the golden bins
    import fastore_amd
    from conftest import manifest, knobs_from_flags, GOLDEN
    out = os.path.join(ROOT, "build", "libsimt_emu_short_list.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DFS_SIMT_EMU", "-DFSM_ITEMS=256", "-shared", "-fPIC", "-o", out,
                           os.path.join(ROOT, "tests", "emu", "mates_simt.cpp"), os.path.join(ROOT, "tests", "emu", "simt.cpp")])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "fastore_amd", "csrc"), "-j", "8", "emu"], stdout=subprocess.DEVNULL)
    code = ("import os, sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import fastore_amd; from conftest import knobs_from_flags, manifest, GOLDEN\n"
            "lib = fastore_amd.load_library(%r)\n"
            "for name, paired, flags in manifest():\n"
            "    if paired:\n"
            "        with fastore_amd.Packer(lib=lib, device_id=0, **knobs_from_flags(flags)) as p: print(p.pe_matcher_check(os.path.join(GOLDEN, name + '.in')))" %
            (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "build", "libfastore_emu.so")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, env=dict(os.environ, FS_EMU_SIMT_MATES=out))      # (the emulation library reads the switch once: a process of its own)
    assert r.returncode == 0, r.stderr
    rows = [eval(l) for l in r.stdout.decode().strip().splitlines()]
    assert rows and all(pairs > 1000 and differing == 0 for pairs, differing in rows), rows


@pytest.mark.parametrize("window", [None, 2, 5, 64, 100])
def test_mate_search_kernel_body_gives_the_host_searchs_rows(simt, monkeypatch, window):
    # fs_match_mates' body (mates_core.h: a wavefront per paired-end bin -- planes from ballots, the sets in an LDS table, candidates
    # priced a lane each) on the emulated wave, through the product's own parity check: every pair of every standard bin of the golden
    # paired-end library, rows against the host's search (LzCompressorPE::CompressPair, FastqCompressor.cpp:4460-4740); small histories
    # make the ring wrap and turn over many times
    import fastore_amd
    from conftest import manifest, knobs_from_flags, GOLDEN
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "fastore_amd", "csrc"), "-j", "8", "emu"], stdout=subprocess.DEVNULL)
    emu = fastore_amd.load_library(os.path.join(ROOT, "build", "libfastore_emu.so"))
    monkeypatch.setenv("FS_EMU_SIMT_MATES", os.path.join(ROOT, "build", "libsimt_emu.so"))
    for name, paired, flags in manifest():
        if not paired:
            continue
        kn = knobs_from_flags(flags)
        if window:
            kn["max_pair_lz_window"] = window
        with fastore_amd.Packer(lib=emu, device_id=0, **kn) as p:
            pairs, differing = p.pe_matcher_check(os.path.join(GOLDEN, name + ".in"))
        assert pairs > 1000 and differing == 0, (name, window, pairs, differing)


@pytest.mark.parametrize("name,paired,flags", manifest())
def test_emission_kernel_body_writes_the_host_walks_streams(simt, name, paired, flags):
    # fs_emit_count / fs_emit_write's body (emit_wave.h: a wavefront per op -- a letter's place is the population count of a ballot, the match
    # bits are a ballot and land in a packed word array) on the lock-step wave emulation: the PRE-ENTROPY bytes of every base-holding stream
    # of every standard bin of every golden library against the host walk writing them itself (the product's own check, fsgpu_emit_check),
    # and the archive.  In a process of its own: the emulation library reads the switch once.
    code = ("import os, sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import fastore_amd; from conftest import knobs_from_flags\n"
            "lib = fastore_amd.load_library(%r)\n"
            "with fastore_amd.Packer(lib=lib, device_id=0, **knobs_from_flags(%r)) as p:\n"
            "    print(p.emit_check(%r)); p.pack_file(%r, sys.argv[1])" %
            (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "build", "libfastore_emu.so"), list(flags), os.path.join(GOLDEN, name + ".in"), os.path.join(GOLDEN, name + ".in")))
    import tempfile
    with tempfile.TemporaryDirectory() as t:
        r = subprocess.run([sys.executable, "-c", code, os.path.join(t, "o")], capture_output=True, env=dict(os.environ, FS_EMU_SIMT_EMIT=os.path.join(ROOT, "build", "libsimt_emu.so")), timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        ops, streams, differing = eval(r.stdout.decode().strip().splitlines()[-1])
        assert ops > 1000 and streams >= 7 * 20 and differing == 0, (ops, streams, differing)
        assert open(os.path.join(t, "o.cdata"), "rb").read() == open(os.path.join(GOLDEN, name + ".ref.cdata"), "rb").read()
