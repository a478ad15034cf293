"""GPU parity tests (MI355X): everything goes through the C ABI of libfastore_amd.so and is compared
bit-for-bit with (a) the oracle's C restatement on seeded inputs, (b) the committed golden archives
made by the real reference, (c) the real reference itself (oracle/_ref) on freshly generated
libraries when its binaries travelled with the checkout."""
import os
import subprocess
import time

import numpy as np
import pytest

from conftest import RefLibs, check_compress_bins_seam, GOLDEN, MODELS, REF_DRIVER, REF_DRIVER_GCC, ROOT, VECTORS, flag_variants, knobs_from_flags, manifest, ref_pipeline, reference_pack, reference_blocks, C1_FLAGS, oracle_ppmd, oracle_qvz, oracle_rc

import sys
sys.path.insert(0, GOLDEN)
import qvz_inputs
from test_host import assert_same_archive

pytestmark = pytest.mark.gpu


GOLDEN_FLAGS = manifest()[0][2]          # the flags six of the seven golden libraries were packed with


@pytest.fixture(scope="module")
def packer():
    """ONE context for the whole module where a test has no reason to make its own: the coder-level tests (the knobs do not reach
    them) and the golden libraries packed with GOLDEN_FLAGS -- a context is a 53 GB arena pool and fourteen lanes, and making one
    per test was a fifth of the suite's time."""
    import fastore_amd
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "fastore_amd", "csrc"), "-j", "8"], stdout=subprocess.DEVNULL)
    p = fastore_amd.Packer(device_id=0, **knobs_from_flags(GOLDEN_FLAGS))
    assert p.device_name.startswith("gfx950"), p.device_name
    yield p
    p.close()


class _Borrowed:
    """`with packer_for(shared, flags) as p`: the module's context when the flags are its own, else a context of the test's own"""
    def __init__(self, shared, flags):
        import fastore_amd
        self.own = None if list(flags) == list(GOLDEN_FLAGS) else fastore_amd.Packer(device_id=0, **knobs_from_flags(flags))
        self.p = self.own or shared
    def __enter__(self):
        self.p.reset_stats()
        return self.p
    def __exit__(self, *a):
        if self.own:
            self.own.close()


def packer_for(shared, flags):
    return _Borrowed(shared, flags)


def test_ppmd_device_matches_reference_vectors(packer):
    names = sorted(f[:-3] for f in os.listdir(VECTORS) if f.startswith("ppmd_") and f.endswith(".in"))
    ins = [open(os.path.join(VECTORS, n + ".in"), "rb").read() for n in names]
    outs = packer.ppmd_encode(ins)
    for n, o in zip(names, outs):
        assert o == open(os.path.join(VECTORS, n + ".out"), "rb").read(), n


def test_rc_device_matches_reference_vectors(packer):
    names = sorted(f[:-3] for f in os.listdir(VECTORS) if f.startswith("rc_") and f.endswith(".in") and f != "rc_empty.in")
    ins = [open(os.path.join(VECTORS, n + ".in"), "rb").read() for n in names]
    outs = packer.rc_encode([MODELS[n[3:]][0] for n in names], ins)
    for n, o in zip(names, outs):
        assert o == open(os.path.join(VECTORS, n + ".out"), "rb").read(), n
    empty = packer.rc_encode(list(range(6)), [b""] * 6)
    assert all(e == open(os.path.join(VECTORS, "rc_empty.out"), "rb").read() for e in empty)


def test_ppmd_device_matches_oracle_many_ragged_streams(packer, oracle):
    rng = np.random.default_rng(101)
    streams = [b"", b"Q"]
    for i in range(300):
        n = int(rng.integers(1, 6000))
        kind = i % 4
        if kind == 0: s = np.clip(38 + np.cumsum(rng.integers(-1, 2, n)), 2, 40).astype(np.uint8)      # quality-like
        elif kind == 1: s = rng.choice(np.frombuffer(b"ACGTN.", dtype=np.uint8), n)                     # hard reads
        elif kind == 2: s = rng.integers(0, 9, n).astype(np.uint8)                                       # flags
        else: s = rng.integers(0, 256, n).astype(np.uint8)                                               # incompressible
        streams.append(s.tobytes())
    got = packer.ppmd_encode(streams)
    assert got[0] == b""
    for s, g in zip(streams[1:], got[1:]):
        assert g == oracle_ppmd(oracle, s)


@pytest.mark.parametrize("waves", [1, 2])
def test_ppmd_device_every_kernel_form_matches_the_oracle(packer, oracle, monkeypatch, waves):
    # the same streams through the one-wave form and the two-wave form (model wave + coder wave): long quality streams with read
    # boundaries (episodes between windows), noise (no windows at all, model restarts), short and ragged streams that
    # follow each other through one workgroup (the hand-over between streams), an empty one
    monkeypatch.setenv("FS_WAVES", str(waves))
    rng = np.random.default_rng(300 + waves)

    def quality(n, read_len):
        steps = np.array([-3, -1, 0, 0, 0, 0, 1, 1])[rng.integers(0, 8, n)].reshape(-1, read_len)
        q = np.empty_like(steps); cur = np.full(steps.shape[0], 38)
        for i in range(read_len):
            cur = np.clip(cur + steps[:, i], 2, 40); q[:, i] = cur
        return q.astype(np.uint8).tobytes()

    streams = [quality(1_500_000, 150), b"", quality(600_000, 100), rng.integers(0, 41, 1_600_000, dtype=np.uint8).tobytes(), b"Q", quality(999 * 37, 37)]
    for i in range(120):
        n = int(rng.integers(1, 9000))
        streams.append(quality(150 * (n // 150 + 1), 150)[:n] if i % 3 else rng.integers(0, 9, n).astype(np.uint8).tobytes())
    # the same stream again and again through one arena: the hint table the stream before left names, at every key, the address
    # the context of that key gets AGAIN (stale hints must be proven wrong by the chain, never believed)
    again = quality(150 * 400, 150)
    streams += [again, again, again]
    got = packer.ppmd_encode(streams)
    for i, (s, g) in enumerate(zip(streams, got)):
        assert g == (oracle_ppmd(oracle, s) if s else b""), (waves, i, len(s))


def test_ppmd_device_model_restart_and_allocator_exhaustion(packer, oracle):
    # > 2 MiB of 41-symbol noise: the text area overruns and the model restarts; 3 MiB of bytes noise
    # additionally drives the sub-allocator through GlueFreeBlocks / AllocUnitsRare
    rng = np.random.default_rng(7)
    a = rng.integers(0, 41, 1_700_000, dtype=np.uint8).tobytes()
    b = rng.integers(0, 256, 2 << 20, dtype=np.uint8).tobytes()
    got = packer.ppmd_encode([a, b])
    assert got[0] == oracle_ppmd(oracle, a)
    assert got[1] == oracle_ppmd(oracle, b)


@pytest.mark.parametrize("windows", ["1", "0", None])
def test_rc_device_matches_oracle_all_models_with_rescale(packer, oracle, monkeypatch, windows):
    # windows: the kernels with the windowed form of the small-alphabet coders (64 symbols per step: rc_core.h) forced on,
    # forced off (the one-symbol loop), and the product's choice (windowed when the launch's range-coded symbols weigh beside its PPMd symbols: here they are all it holds)
    if windows is not None:
        monkeypatch.setenv("FS_RC_WINDOWS", windows)
    rng = np.random.default_rng(33)
    models, ins = [], []
    for name, (mid, bits, order, adv) in MODELS.items():
        A = 1 << bits
        for n in (1, 17, 5000, 120000):                       # 120000 symbols force TSymbolCoderRC::Rescale on hot models
            sym = np.minimum(rng.geometric(0.3, n) - 1, A - 1).astype(np.uint8) if A > 2 else rng.integers(0, 2, n, dtype=np.uint8)
            ctx = rng.integers(0, min(A, 8), n, dtype=np.uint8)
            pairs = np.stack([sym, ctx], 1).tobytes()
            models.append(mid); ins.append((name, pairs))
    # quality-like streams: runs of one symbol in position-dependent contexts -- most of a window on one row, rescales in the middle of windows
    for name in ("a8o6", "a2o10", "s2o4"):
        mid, bits, order, adv = MODELS[name]
        A = 1 << bits; n = 300_000
        sym = np.clip(np.cumsum(rng.integers(-1, 2, n)) // 60 % A, 0, A - 1).astype(np.uint8)
        ctx = ((np.arange(n) % 150) * A // 150).astype(np.uint8) if adv else np.zeros(n, np.uint8)
        models.append(mid); ins.append((name, np.stack([sym, ctx], 1).tobytes()))
    got = packer.rc_encode(models, [p for _, p in ins])
    for (name, pairs), g in zip(ins, got):
        assert g == oracle_rc(oracle, name, pairs), name
    # model 6 = <256,1> with a 6-bit ctx0 slot (dense table for read-id streams): same bytes as model 5
    dense = [(n, p) for n, p in ins if n == "a256o1"]
    got6 = packer.rc_encode([6] * len(dense), [p for _, p in dense])
    for (name, pairs), g in zip(dense, got6):
        assert g == oracle_rc(oracle, name, pairs)


def test_qvz_device_matches_reference_vectors(packer):
    for name in qvz_inputs.CASES:                        # incl. the count rescaling and the > 64-symbol alphabets
        lens, quals = qvz_inputs.reads_case(name)
        got = packer.qvz_encode(qvz_inputs.footer_for(name), [(lens, quals)])[0]
        assert got == open(os.path.join(VECTORS, name + ".out"), "rb").read(), name


def test_qvz_device_matches_oracle_many_ragged_blocks(packer, oracle):
    rng = np.random.default_rng(303)
    for footer, top, maxlen in ((qvz_inputs.qvz_footer(), 42, 60), (qvz_inputs.wide_footer(), 72, 3)):
        blocks = []
        for i in range(200):
            lens = rng.integers(1, maxlen + 1, int(rng.integers(1, 400))).astype(np.uint32)
            blocks.append((lens, rng.integers(2 if top == 42 else 0, top, int(lens.sum())).astype(np.uint8)))
        blocks.append((np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.uint8)))          # empty block: just the coder's tail
        got = packer.qvz_encode(footer, blocks)
        for (lens, quals), g in zip(blocks, got):
            assert g == oracle_qvz(oracle, footer, lens, quals)


def test_qvz_long_block_through_the_coder_wave_and_through_one_wave(monkeypatch, oracle):
    # a block long enough for the launch to take the two-wave kernel: the symbols' counts go through the ring, fractions and the interval's pass run on
    # the coder wave (fsppmd::coder_wave<true, fsqvz::WaveCoder>); beside it short blocks (other streams of the same launch share the coder wave one
    # after the other) and the same launch forced onto the one-wave kernel: all equal to the oracle's coder (arith.cpp / qv_compressor.cpp restated)
    import fastore_amd
    rng = np.random.default_rng(909)
    footer = qvz_inputs.qvz_footer()
    reads = 5200                                             # x 60 scores = 312 000 symbols >= 256 Ki
    lens = np.full(reads, 60, dtype=np.uint32)
    walk = np.clip(38 + np.cumsum(rng.integers(-1, 2, reads * 60)) % 30, 2, 41).astype(np.uint8)
    blocks = [(lens, walk)]
    for i in range(20):
        l2 = rng.integers(1, 61, int(rng.integers(1, 300))).astype(np.uint32)
        blocks.append((l2, rng.integers(2, 42, int(l2.sum())).astype(np.uint8)))
    want = [oracle_qvz(oracle, footer, l, q) for l, q in blocks]
    for waves in (None, "1", "2"):
        if waves is None:
            monkeypatch.delenv("FS_WAVES", raising=False)
        else:
            monkeypatch.setenv("FS_WAVES", waves)
        with fastore_amd.Packer(device_id=0) as p:
            got = p.qvz_encode(footer, blocks)
        assert got == want, waves


def test_qvz_device_rejects_bad_input(packer):
    import fastore_amd
    f = qvz_inputs.qvz_footer()
    with pytest.raises(fastore_amd.FastoreError):
        packer.qvz_encode(f[:1000], [(np.array([1], dtype=np.uint32), np.array([3], dtype=np.uint8))])      # truncated codebook
    with pytest.raises(fastore_amd.FastoreError):
        packer.qvz_encode(f, [(np.array([1], dtype=np.uint32), np.array([99], dtype=np.uint8))])             # value outside the alphabet
    with pytest.raises(fastore_amd.FastoreError):
        packer.qvz_encode(f, [(np.array([61], dtype=np.uint32), np.full(61, 30, dtype=np.uint8))])           # read longer than the codebook


@pytest.mark.parametrize("name,paired,flags", manifest())
def test_gpu_pack_reproduces_reference_archives(tmp_path, packer, name, paired, flags):
    with packer_for(packer, flags) as p:
        st = p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / "o"))
        assert st["encode_kernel_ms"] > 0 and st["stream_items"] >= 15 * st["bins"]
    assert_same_archive(str(tmp_path / "o"), os.path.join(GOLDEN, name + ".ref"))


def test_cli_is_a_drop_in_for_fastore_pack_e(tmp_path):
    cli = os.path.join(ROOT, "fastore_amd", "fastore_pack")
    name, paired, flags = manifest()[0]
    r = subprocess.run([cli, "e", "-i" + os.path.join(GOLDEN, name + ".in"), "-o" + str(tmp_path / "o"), "-t4", "-v"] + flags, capture_output=True)
    assert r.returncode == 0, r.stderr
    assert b"Parts processed" in r.stderr
    assert open(os.path.join(GOLDEN, name + ".ref.vout"), "rb").read() in r.stdout      # the reference's -v statistics
    assert_same_archive(str(tmp_path / "o"), os.path.join(GOLDEN, name + ".ref"))


@pytest.mark.skipif(not (os.path.exists(REF_DRIVER) and os.path.exists(REF_DRIVER_GCC)), reason="reference binaries (oracle/_ref) not shipped")
# (single-end --reduced and --lossy: the golden libraries se_reduced / se_qvz and the bench's legs; paired-end in every mode here)
@pytest.mark.parametrize("paired,q,reads", [(False, 0, 60000), (True, 0, 40000), (True, 2, 40000), (True, 3, 40000)])
def test_gpu_pack_equals_live_reference_on_fresh_library(tmp_path, paired, q, reads):
    import fastore_amd
    t = str(tmp_path)
    binned, pe = ref_pipeline(t, "lib", reads, 150, reads * 150 // 50, 77 + q + int(paired), paired, q)
    reference_pack(binned, os.path.join(t, "ref"), pe)
    with fastore_amd.Packer(device_id=0) as p:
        st = p.pack_file(binned, os.path.join(t, "gpu"))
    assert_same_archive(os.path.join(t, "gpu"), os.path.join(t, "ref"))
    assert st["bins"] >= 5
    # independent check: the reference DECODER accepts our archive and returns the same multiset of reads
    outs = [os.path.join(t, "dec_1.fastq")] + ([os.path.join(t, "dec_2.fastq")] if paired else [])
    subprocess.check_call([REF_DRIVER, "unpack", "-i" + os.path.join(t, "gpu"), "-o" + " ".join(outs), "-t1"] + pe, timeout=60)
    def records(path):
        lines = open(path, "rb").read().split(b"\n")
        return sorted(zip(lines[0::4], lines[1::4], lines[3::4]))
    if q == 0:
        assert records(outs[0]) == records(os.path.join(t, "lib_1.fastq"))
        if paired:
            assert records(outs[1]) == records(os.path.join(t, "lib_2.fastq"))


def test_gpu_pack_is_deterministic_and_sizes_add_up(tmp_path, monkeypatch):
    import fastore_amd
    name, paired, flags = manifest()[1]
    outs = []
    tails = []
    # wave counts and pipeline slicing change scheduling, never the bytes.  (16, 6) and (300, 3): pools of 2 and 38 arena slots per XCD
    # under several launches at once -- workgroups that find their XCD's slots taken leave without waiting, and a launch none of whose
    # workgroups found one is made again (coder_tail_launches)
    for i, (waves, slices) in enumerate(((0, 1), (37, 1), (0, 6), (300, 3), (16, 6))):
        with fastore_amd.Packer(device_id=0, max_waves=waves, pipeline_slices=slices, **knobs_from_flags(flags)) as p:
            st = p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / ("o%d" % i)))
            tails.append(p.stats()["coder_tail_launches"])
        outs.append(open(str(tmp_path / ("o%d.cdata" % i)), "rb").read())
    # ... and a workgroup's share of one launch cut to a stream or two (FS_WG_BUDGET): every launch is made again and again for what its
    # workgroups left in the queue -- the same bytes
    monkeypatch.setenv("FS_WG_BUDGET", "6000")
    with fastore_amd.Packer(device_id=0, max_waves=37, pipeline_slices=1, **knobs_from_flags(flags)) as p:      # (37 workgroups for all the streams: alone they need no second launch)
        p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / "ob"))
        tails.append(p.stats()["coder_tail_launches"])
    monkeypatch.delenv("FS_WG_BUDGET")
    outs.append(open(str(tmp_path / "ob.cdata"), "rb").read())
    assert tails[-1] > 0, tails
    print("coder tail launches per configuration:", tails)
    assert all(o == outs[0] for o in outs[1:])
    assert outs[0] == open(os.path.join(GOLDEN, name + ".ref.cdata"), "rb").read()


@pytest.mark.parametrize("name,paired,sha,flags", flag_variants())
def test_gpu_pack_under_non_default_flags(tmp_path, name, paired, sha, flags):
    import fastore_amd, hashlib
    with fastore_amd.Packer(device_id=0, **knobs_from_flags(flags)) as p:
        p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / "o"))
    assert hashlib.sha256(open(str(tmp_path / "o.cdata"), "rb").read()).hexdigest() == sha


@pytest.mark.parametrize("name,paired,flags", manifest())
def test_gpu_compress_bins_seam(packer, name, paired, flags):
    # fsgpu_compress_bins: unpacked standard bins in, the reference's blocks out (bin by bin)
    import fastore_amd
    with packer_for(packer, flags) as p:
        check_compress_bins_seam(fastore_amd, p, name, flags)


def test_rc_device_rejects_symbols_outside_the_alphabet(packer):
    # a symbol or context a model has no statistic for must end the stream with an error, not spin in the coder's
    # normalisation loop (frequency 0) or index past its table
    import fastore_amd
    ok = bytes([1, 0, 3, 1, 7, 2])
    assert len(packer.rc_encode([MODELS["a8o4"][0]], [ok])[0]) >= 8
    for name, bad in (("s2o4", bytes([2, 0])), ("a8o4", bytes([1, 0, 9, 0])), ("a8o6", bytes([0, 0, 200, 1])), ("a2o10", bytes([1, 0, 0, 3]))):
        with pytest.raises(fastore_amd.FastoreError, match="alphabet"):
            packer.rc_encode([MODELS[name][0]], [bad])
    with pytest.raises(fastore_amd.FastoreError, match="alphabet"):
        packer.rc_encode([6], [bytes([65, 3, 66, 64])])          # model 6: ctx0 must be < 64


def test_gpu_libraries_larger_than_a_device_batch(tmp_path):
    # several device batches per library (and libraries sharing batches): same bytes
    import fastore_amd
    fx = manifest()[:2]
    with fastore_amd.Packer(device_id=0, batch_bases=300_000, **knobs_from_flags(fx[0][2])) as p:
        ins = [os.path.join(GOLDEN, name + ".in") for name, _, _ in fx]
        outs = [str(tmp_path / ("m_" + name)) for name, _, _ in fx]
        p.pack_files(ins, outs)
        for (name, _, _), o in zip(fx, outs):
            assert open(o + ".cdata", "rb").read() == open(os.path.join(GOLDEN, name + ".ref.cdata"), "rb").read()


def _archive_signatures(prefix):
    import struct
    m = open(prefix + ".cmeta", "rb").read(); foff, _ = struct.unpack_from("<QQ", m, 0); n, = struct.unpack_from("<I", m, foff)
    return list(struct.unpack_from("<%dI" % n, m, foff + 4 + 8 * n))


@pytest.mark.parametrize("lib", ["se_long", "pe_long"])
def test_gpu_pack_equals_live_reference_on_a_library_with_long_streams(tmp_path, ref_libs, lib):
    # BASELINE-shaped bins from a small library (conftest.RefLibs: a 6 kbp genome at 10 000-fold coverage): standard bins of > 25 000
    # reads (quality streams of > 3.5 M PPMd symbols) single-end, > 8 000 pairs (both mates in one stream: > 2.4 M symbols)
    # paired-end -- the tail of a device step.  The reference packs with one worker, so the archives are compared whole.
    import fastore_amd, struct
    binned, pe, _ = ref_libs.library(lib)
    ref, _ = ref_libs.packed(lib)
    t = str(tmp_path)
    with fastore_amd.Packer(device_id=0) as p:
        st = p.pack_file(binned, os.path.join(t, "gpu"))
    assert_same_archive(os.path.join(t, "gpu"), ref)
    got = reference_blocks(os.path.join(t, "gpu"))
    sigs = _archive_signatures(os.path.join(t, "gpu"))
    assert sigs[0] == max(sigs) and sigs[1:] == sorted(sigs[1:])
    # the longest quality stream of the library: records of the largest standard block x 150 (x 2 mates)
    biggest = max(struct.unpack_from(">Q", b, 4)[0] for sg, b in got.items() if sg != max(sigs))
    assert biggest * 150 * (2 if pe else 1) > (2 << 20), biggest
    assert st["host_coded_symbols"] == 0          # every standard-bin stream above was coded on the device
    # the same library through the CLI: a ONE-SHOT context (lanes and staging buffers made beside the front end, matcher
    # lanes by their threads, pageable staging, archive pages reserved ahead, no teardown) must write the same archive
    subprocess.check_call([fastore_amd.PACK_CLI, "e", "-i" + binned, "-o" + os.path.join(t, "cli")] + C1_FLAGS + pe, timeout=90)
    for e in (".cdata", ".cmeta"):
        assert open(os.path.join(t, "cli" + e), "rb").read() == open(os.path.join(t, "gpu" + e), "rb").read(), e


@pytest.mark.parametrize("name,paired,flags", manifest()[:2])
def test_gpu_library_of_several_batches_goes_through_two_pipelines(tmp_path, monkeypatch, name, paired, flags):
    # fsgpu_pack_file on a library of more bases than a device batch holds: two pipelines on the one device (the heaviest bins /
    # all the others), or one (FS_SPLIT_PIPELINES=0) -- the reference's archive either way
    import fastore_amd
    ref = open(os.path.join(GOLDEN, name + ".ref.cdata"), "rb").read()
    for split in ("1", "0"):
        monkeypatch.setenv("FS_SPLIT_PIPELINES", split)
        with fastore_amd.Packer(device_id=0, batch_bases=300_000, **knobs_from_flags(flags)) as p:
            st = p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / ("s" + split)))
        assert st["device_batches"] >= 2
        assert open(str(tmp_path / ("s" + split)) + ".cdata", "rb").read() == ref, split


def test_gpu_pack_many_batches_and_model_restarts_inside_standard_bins(tmp_path, ref_libs):
    # What a library of BASELINE configs[2]'s size does to the pipeline, at a size a test can afford (conftest.RefLibs "pe_noisy":
    # 70 000 pairs over a 1.5 kbp genome): a bin of 17 000 pairs, a device batch budget that cuts the standard bins into >= 3
    # batches (blocks of the earlier batches wait behind block 0), and quality scores without structure (gen_fastq
    # --noisy-quality), so that the PPMd model of a standard bin's quality stream outgrows its heap and restarts >= 3 times inside
    # ONE stream (ppmd/Model.cpp:109-140 + the sub-allocator's exhaustion paths, SubAlloc.hpp:98-163).  The whole archive against
    # the live reference's.
    import fastore_amd
    binned, pe, _ = ref_libs.library("pe_noisy")
    ref, _ = ref_libs.packed("pe_noisy")
    t = str(tmp_path)
    with fastore_amd.Packer(device_id=0, batch_bases=7_000_000) as p:
        st = p.pack_file(binned, os.path.join(t, "gpu"))
    assert_same_archive(os.path.join(t, "gpu"), ref)
    assert st["device_batches"] >= 3, st["device_batches"]
    assert st["ppmd_max_restarts"] >= 3, st["ppmd_max_restarts"]
    assert st["host_coded_symbols"] == 0
    sigs = _archive_signatures(os.path.join(t, "gpu"))
    assert sigs[0] == max(sigs) and sigs[1:] == sorted(sigs[1:])          # -t1 order although the batches finished one after the other


def test_gpu_rank_sharded_pack_on_one_device(tmp_path):
    # the multi-GPU path (fsgpu_config.rank/world_size + fsgpu_merge_parts), both ranks on device 0 one after the other:
    # the merged archive must be the single-writer archive, for SE and PE and an odd world size
    import fastore_amd
    for (name, paired, flags), world in zip(manifest()[:2], (2, 3)):
        out = str(tmp_path / ("s_" + name))
        records = 0
        for r in range(world):
            with fastore_amd.Packer(device_id=0, rank=r, world_size=world, **knobs_from_flags(flags)) as p:
                st = p.pack_file(os.path.join(GOLDEN, name + ".in"), out)
                records += st["records"] + st["block0_records"]
                assert st["bins"] > 0
        fastore_amd.merge_parts(out, world)
        assert_same_archive(out, os.path.join(GOLDEN, name + ".ref"))
        assert not [f for f in os.listdir(tmp_path) if ".part" in f]


def test_gpu_shard_api_on_one_device(tmp_path):
    # the multi-GPU path proper (fsgpu_shard_pack / _table / _write: LPT shares, summed size table, positional writes),
    # three ranks on device 0 one after the other
    import fastore_amd
    name, paired, flags = manifest()[1]
    out = str(tmp_path / "o")
    world = 3
    packers = [fastore_amd.Packer(device_id=0, rank=r, world_size=world, **knobs_from_flags(flags)) for r in range(world)]
    try:
        tables = [p.shard_pack(os.path.join(GOLDEN, name + ".in")) for p in packers]
        total = np.stack([t[1] for t in tables]).sum(axis=0)
        for p in packers:
            p.shard_write(out, total)
    finally:
        for p in packers:
            p.close()
    assert_same_archive(out, os.path.join(GOLDEN, name + ".ref"))


def test_gpu_cli_two_contexts_on_one_box(tmp_path):
    # `fastore_pack e -G2` needs two devices; on a one-GPU box the same code path is driven with -R/-N + merge
    import fastore_amd
    cli = os.path.join(ROOT, "fastore_amd", "fastore_pack")
    name, paired, flags = manifest()[0]
    lib = fastore_amd.load_library()
    out = str(tmp_path / "o")
    if lib.fsgpu_device_count() >= 2:
        r = subprocess.run([cli, "e", "-i" + os.path.join(GOLDEN, name + ".in"), "-o" + out, "-G2", "-v"] + flags, capture_output=True)
        assert r.returncode == 0, r.stderr
        assert open(os.path.join(GOLDEN, name + ".ref.vout"), "rb").read() in r.stdout
    else:
        for r in range(2):
            assert subprocess.run([cli, "e", "-i" + os.path.join(GOLDEN, name + ".in"), "-o" + out, "-R%d" % r, "-N2"] + flags).returncode == 0
        fastore_amd.merge_parts(out, 2)
    assert_same_archive(out, os.path.join(GOLDEN, name + ".ref"))


def test_gather_quality_device_matches_the_restated_unpack(packer):
    # fs_gather_quality through the C ABI: ragged strings (0 .. 310 scores), unaligned bit offsets, both orientations,
    # emitted in a shuffled order -- against the numpy restatement of the reference's unpack + CompressReadQuality
    from conftest import quality_gather_case
    for seed in (1, 2, 3, 4):
        packed, strings, expect = quality_gather_case(seed, n_strings=3000)
        assert packer.gather_quality(packed, strings) == expect
    st = packer.stats()
    assert st["gather_symbols"] > 0 and st["gather_kernel_ms"] > 0


def test_gather_quality_pairs_device_matches_the_restated_symbolisation(packer):
    # fs_gather_quality_pairs (8-bin and binary archives): 3- and 1-bit scores, 'N' positions left out, contexts by emitted
    # index, thresholds that map the stored bit either way -- against the numpy restatement
    from conftest import quality_gather_binned_case
    for bits, thr in ((3, 20), (1, 20), (1, 5), (1, 41)):
        packed, strings, expect = quality_gather_binned_case(100 + bits + thr, bits, thr, n_strings=2000)
        assert packer.gather_quality_binned(packed, bits, thr, strings) == expect, (bits, thr)


@pytest.mark.parametrize("which", [0, 2, 4, 5])
def test_gpu_quality_streams_come_from_the_device_gather(tmp_path, monkeypatch, which):
    # a library packed from .b* files: the scores go to the device packed (six, three or one bit each) and the quality
    # streams are built there -- bytes for PPMd (lossless), (symbol, context) pairs (8-bin, binary), and for --lossy the
    # (quantizer, state) words of the QVZ coder, with the quantizer chosen per score from the archive's codebook and the
    # draws of its WELL generator (fs_gather_quality_qvz); the host symbolisation (FS_DEVICE_QUALITY=0) gives the same archive
    # with more H2D bytes
    import fastore_amd
    name, paired, flags = manifest()[which]              # lossless, 8-bin, binary, QVZ
    ref = open(os.path.join(GOLDEN, name + ".ref.cdata"), "rb").read()
    h2d = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FS_DEVICE_QUALITY", mode)
        with fastore_amd.Packer(device_id=0, **knobs_from_flags(flags)) as p:
            st = p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / ("o" + mode)))
        assert open(str(tmp_path / ("o" + mode)) + ".cdata", "rb").read() == ref
        h2d[mode] = st["h2d_bytes"]
        assert (st["gather_symbols"] > 0) == (mode == "1")
    # (--lossy: the codebook's quantizer tables travel with every slice -- 144 bytes per conditional quantizer, more than this small
    # library's scores; they pay from a few million scores on: bench.py's `lossy` leg)
    assert h2d["1"] < h2d["0"] or which == 5


@pytest.mark.parametrize("name,paired,flags", manifest())
def test_device_matcher_agrees_with_the_host_window_scan(packer, name, paired, flags):
    # matcher.hip against the host's serial restatement of ReadsClassifierSE::ConstructMatchTree's window search, read
    # by read over every match-tree construction (top level + stored sub-trees) of every golden bin: matched read,
    # cost, shift, mismatch-free flag, exact-duplicate flag
    with packer_for(packer, flags) as p:
        reads, differing = p.matcher_check(os.path.join(GOLDEN, name + ".in"))
    assert reads > 1000 and differing == 0, (reads, differing)


@pytest.mark.parametrize("name,paired,flags", [m for m in manifest() if m[1]])
def test_device_mate_search_agrees_with_the_host_search(name, paired, flags):
    # fs_match_mates (paired-end bins: LzCompressorPE::CompressPair's history search, FastqCompressor.cpp:4610-4959) row for row
    # against the host's serial search on every standard bin of the golden paired-end libraries; small histories too
    import fastore_amd
    for window in (None, 2, 5, 64, 1000):
        kn = knobs_from_flags(flags)
        if window:
            kn["max_pair_lz_window"] = window
        with fastore_amd.Packer(device_id=0, **kn) as p:
            pairs, differing = p.pe_matcher_check(os.path.join(GOLDEN, name + ".in"))
        assert pairs > 1000 and differing == 0, (window, pairs, differing)


def test_device_mate_search_on_a_fresh_library_and_same_archive_either_way(tmp_path, monkeypatch, ref_libs):
    # a fresh 200 000-pair library (bins of thousands of pairs: the history is full and turns over many times): rows against the
    # host's search, and the archive with the device's searches and with the host's against the live reference
    import fastore_amd
    binned, pe, _ = ref_libs.library("pe_long")
    ref, _ = ref_libs.packed("pe_long")
    t = str(tmp_path)
    with fastore_amd.Packer(device_id=0) as p:
        pairs, differing = p.pe_matcher_check(binned)
        assert pairs > 50_000 and differing == 0, (pairs, differing)
    # (the archive with the host's searches is test_gpu_pack_equals_live_reference_on_a_library_with_long_streams[pe_long])
    # the device's searches handed over and run in batches of bins
    for mode in ("2",):
        monkeypatch.setenv("FS_DEVICE_MATES", mode)
        with fastore_amd.Packer(device_id=0) as p:
            st = p.pack_file(binned, os.path.join(t, "gpu" + mode))
        assert_same_archive(os.path.join(t, "gpu" + mode), ref)
        assert st["mate_pairs"] > 50_000 and st["mate_kernel_ms"] > 0, (mode, st["mate_pairs"])


@pytest.mark.parametrize("name,paired,flags", manifest())
def test_device_unpack_of_the_bases_agrees_with_the_host_unpack(tmp_path, monkeypatch, packer, name, paired, flags):
    # fs_unpack_planes (the reader of FastqPacker.cpp:290-411 on the device): the window search's bit planes from
    # the bin's .bdna bytes -- two- and three-bit reads, the signature that is not stored, sub-tree reads with their own
    # signature, exact-match records that have no bases of their own -- word by word against the planes fs_pack_bases makes
    # from the host's unpacked bases, on every golden bin; the rows of the search on them against the host scan; and the
    # archive, which is the reference's either way while only the packed bytes go up
    import fastore_amd
    with packer_for(packer, flags) as p:
        words, differing, reads, rows = p.unpack_check(os.path.join(GOLDEN, name + ".in"))
    assert words >= 40 * reads and differing == 0 and reads > 1000 and rows == 0, (words, differing, reads, rows)
    ref = open(os.path.join(GOLDEN, name + ".ref.cdata"), "rb").read()
    seen = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FS_DEVICE_UNPACK", mode)
        with fastore_amd.Packer(device_id=0, **knobs_from_flags(flags)) as p:
            seen[mode] = p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / ("o" + mode)))
        assert open(str(tmp_path / ("o" + mode)) + ".cdata", "rb").read() == ref
    assert seen["0"]["matcher_unpacked_reads"] == 0 and seen["0"]["matcher_reads"] > 0
    assert seen["1"]["matcher_unpacked_reads"] == seen["1"]["matcher_reads"] > 0
    assert seen["1"]["matcher_bases_h2d_bytes"] < 0.45 * seen["0"]["matcher_bases_h2d_bytes"]


@pytest.mark.parametrize("window", [2, 5, 64, 65, 129, 1025])
def test_device_matcher_window_sizes(window):
    # windows of one slot up to the largest the kernel takes (one thread per slot, 1 .. 16 wavefronts): ring wrap-around,
    # dummy slots while the window fills, duplicates that stay out of it
    import fastore_amd
    name, paired, flags = manifest()[0]
    kn = knobs_from_flags(flags); kn["max_lz_window"] = window
    with fastore_amd.Packer(device_id=0, **kn) as p:
        reads, differing = p.matcher_check(os.path.join(GOLDEN, name + ".in"))
    assert reads > 1000 and differing == 0, (window, reads, differing)


@pytest.mark.parametrize("lib", ["se_long", "pe_long"])
def test_device_matcher_on_fresh_libraries_and_same_archive_either_way(tmp_path, monkeypatch, ref_libs, lib):
    # C1-profile libraries (bins of thousands of reads: full 1023-slot windows, sub-trees): the device's rows equal the host
    # scan's, and the archive is the same bytes with the device matcher and with the host scan (FS_DEVICE_MATCHER=0)
    import fastore_amd
    binned, pe, _ = ref_libs.library(lib)
    reads = RefLibs.shapes[lib][0]
    t = str(tmp_path)
    kn = dict(min_bin_size=256, max_lz_window=1024, max_pair_lz_window=1024, extra_reduce_hard_reads=1, min_consensus_size=10, max_hamming_distance=8)
    with fastore_amd.Packer(device_id=0, **kn) as p:
        n, differing = p.matcher_check(binned)
        assert n > reads // 4 and differing == 0, (n, differing)
        words, wrong, n2, rows = p.unpack_check(binned)         # ... and the planes from the packed bases against those from the unpacked ones
        assert words >= 40 * n2 and wrong == 0 and n2 == n and rows == 0, (words, wrong, n2, rows)
        ids, bad_bins = p.tokeniser_check(binned)              # the same libraries through the device tokeniser
        assert ids > reads // 4 and bad_bins == 0, (ids, bad_bins)
        ops, streams, wrong = p.emit_check(binned)             # ... and through the emission kernels: bins of tens of thousands of ops, contigs, sub-trees, long runs
        assert ops > reads // 4 and wrong == 0, (ops, streams, wrong)
        p.pack_file(binned, os.path.join(t, "dev"))
    # the lighter bins' rule -- on the device while fewer than n threads are inside a search, else the host's scan -- for EVERY bin: which
    # bins go where changes from run to run, the archive does not
    monkeypatch.setenv("FS_MATCHER_BINS", "0"); monkeypatch.setenv("FS_SEARCH_SURPLUS", "2")
    with fastore_amd.Packer(device_id=0, **kn) as p:
        st = p.pack_file(binned, os.path.join(t, "gate"))
    assert 0 < st["matcher_reads"], st["matcher_reads"]
    assert open(os.path.join(t, "dev.cdata"), "rb").read() == open(os.path.join(t, "gate.cdata"), "rb").read()
    monkeypatch.delenv("FS_MATCHER_BINS"); monkeypatch.delenv("FS_SEARCH_SURPLUS")
    monkeypatch.setenv("FS_DEVICE_MATCHER", "0"); monkeypatch.setenv("FS_DEVICE_IDS", "0"); monkeypatch.setenv("FS_DEVICE_QUALITY", "0")
    with fastore_amd.Packer(device_id=0, **kn) as p:
        p.pack_file(binned, os.path.join(t, "host"))
    assert open(os.path.join(t, "dev.cdata"), "rb").read() == open(os.path.join(t, "host.cdata"), "rb").read()


@pytest.mark.parametrize("which", [0, 1])
def test_gpu_read_id_streams_come_from_the_device_tokeniser(tmp_path, monkeypatch, which):
    # fs_tokenise_ids: the IdToken / IdValue streams written on the device from the packed headers (SE and PE golden
    # libraries: a constant token, numeric fields, the pair field) -- the archive is the reference's either way
    import fastore_amd
    name, paired, flags = manifest()[which]
    ref = open(os.path.join(GOLDEN, name + ".ref.cdata"), "rb").read()
    for mode in ("1", "0"):
        monkeypatch.setenv("FS_DEVICE_IDS", mode)
        with fastore_amd.Packer(device_id=0, **knobs_from_flags(flags)) as p:
            st = p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / ("o" + mode)))
        assert open(str(tmp_path / ("o" + mode)) + ".cdata", "rb").read() == ref
        assert st["tokenised_ids"] == (st["records"] if mode == "1" else 0)


def test_device_tokeniser_agrees_with_the_host_tokeniser(packer):
    # fs_tokenise_ids against the host restatement of IHeaderStoreBase::CompressReadId, bin by bin, on every golden library with
    # read ids (SE, PE with the pair field, 8-bin, QVZ, bin-stage flavour)
    total = 0
    for name, paired, flags in manifest():
        with packer_for(packer, flags) as p:
            ids, differing = p.tokeniser_check(os.path.join(GOLDEN, name + ".in"))
        assert differing == 0, (name, ids, differing)
        total += ids
    assert total > 10000


@pytest.mark.parametrize("name,paired,flags", manifest())
def test_device_emission_agrees_with_the_host_walk(tmp_path, monkeypatch, packer, name, paired, flags):
    # fs_emit_count / fs_emit_scan / fs_emit_write, fs_rle_binary, fs_rle0 (SURVEY 8 a6 + a11): the streams that hold bases --
    # HardReads, LettersX, Match, MatchBinary, CMatch, CLetters, their paired-end counterparts -- and the run-length coded LZ ids,
    # written on the device from the ops the walk leaves: PRE-ENTROPY bytes of every standard bin of every golden library against
    # the host's walk writing them itself (CompressHardRead ... StoreContigDefinition, BinaryRleEncoder, Rle0Encoder), and the
    # archive with the kernels (the default) and without (FS_DEVICE_EMIT=0)
    import fastore_amd
    with packer_for(packer, flags) as p:
        ops, streams, differing = p.emit_check(os.path.join(GOLDEN, name + ".in"))
    assert ops > 1000 and streams >= 7 * 20 and differing == 0, (ops, streams, differing)
    ref = open(os.path.join(GOLDEN, name + ".ref.cdata"), "rb").read()
    monkeypatch.setenv("FS_DEVICE_EMIT", "0")
    with fastore_amd.Packer(device_id=0, **knobs_from_flags(flags)) as p:
        p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / "o0"))
    assert open(str(tmp_path / "o0.cdata"), "rb").read() == ref


def test_gpu_shard_set_api_on_one_device(tmp_path):
    # the N > 1 bench path (fsgpu_shard_pack_set / _table_of / _write_of): a SET of libraries -- SE lossless, SE bin-stage
    # flavour, SE lossless again -- bin-sharded over three ranks that share device 0, every rank's share of all three in one
    # device pipeline; summed size tables, positional writes; every archive must be the single-writer archive
    import fastore_amd
    names = ["se_lossless", "se_c0", "se_lossless"]
    flags = [m for m in manifest() if m[0] == "se_lossless"][0][2]
    ins = [os.path.join(GOLDEN, n + ".in") for n in names]; outs = [str(tmp_path / ("o%d" % i)) for i in range(len(names))]
    world = 3
    packers = [fastore_amd.Packer(device_id=0, rank=r, world_size=world, **knobs_from_flags(flags)) for r in range(world)]
    try:
        tables = [p.shard_pack_set(ins) for p in packers]
        for i in range(len(names)):
            total = np.stack([t[i][1] for t in tables]).sum(axis=0)
            for p in packers:
                p.shard_write_of(i, outs[i], total)
    finally:
        for p in packers:
            p.close()
    for i, n in enumerate(names):
        assert_same_archive(outs[i], os.path.join(GOLDEN, n + ".ref"))


@pytest.mark.skipif(not (os.path.exists(REF_DRIVER) and os.path.exists(REF_DRIVER_GCC)), reason="reference binaries (oracle/_ref) not shipped")
@pytest.mark.timeout(170)
def test_bench_two_ranks_rehearsed_on_one_device_against_the_live_reference(tmp_path):
    # the N > 1 bench path end to end, every round: two ranks (torch.distributed.run, one process each) share device 0 and talk over
    # gloo (`--rehearse`; on a node the same code runs over RCCL): a SET of two small libraries bin-sharded over the ranks in one
    # device pipeline, the one all-reduce of the size tables, positional writes -- and both archives, and the strong line's,
    # block for block against the live reference's pack of each library
    import json, sys
    port = 29600 + os.getpid() % 300
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--reads", "100000", "--steps", "1", "--warmup", "1", "--work", str(tmp_path)],
                       capture_output=True, timeout=150, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    # the headline: the ONE library bin-sharded over the ranks (strong scaling, BASELINE.json's configs[3] / [4]); beside it the SET of two
    assert res["n_gpus"] == 2 and res["ranks"] == 2 and res["scaling"] == "strong" and res["collective_backend"].startswith("gloo")
    assert res["value"] > 0 and "ONE library" in res["config"]["workload"]
    assert res["parity"]["every_block_bit_identical_to_reference"], res["parity"]
    ws = res["parity"]["weak_set"]
    assert ws["every_library_every_block_bit_identical_to_reference"] and ws["per_library"] == [True, True], res["parity"]
    assert res["weak_set"]["scaling"] == "weak" and res["weak_set"]["value"] > 0


@pytest.mark.timeout(400, method="thread")
def test_two_cli_processes_share_the_device_with_long_streams_and_both_end(tmp_path, ref_libs):
    # Two `fastore_pack e` processes at once on the one device, DEFAULT environment; FS_TWO_PROC_ROUNDS rounds (8 in the suite, which must end
    # inside the driver's window -- a round beside another process takes 1.4-5.5 s; the twenty-round runs are profiles/r05_cli_two_processes.txt).  Rounds 1-4 had a workgroup without an
    # arena slot spin inside its kernel for a workgroup of ANOTHER launch to free one; with two processes' hardware queues competing that
    # was a standstill (profiles/r04_cli_two_processes.txt).  No workgroup waits for another any more (engine.hip: a full partition is left
    # at once, the host launches again if nobody took the streams).  Each child has 120 s; a child that hangs is killed and fails the test.
    import fastore_amd
    binned, pe, _ = ref_libs.library("se_long")
    ref, _ = ref_libs.packed("se_long")
    t = str(tmp_path)
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES",)}
    want = {e: open(ref + e, "rb").read() for e in (".cdata",)}
    times = []
    for rnd in range(int(os.environ.get("FS_TWO_PROC_ROUNDS", "8"))):
        t0 = time.monotonic()
        kids = [subprocess.Popen([fastore_amd.PACK_CLI, "e", "-i" + binned, "-o" + os.path.join(t, "o%d" % k)] + C1_FLAGS + pe, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE) for k in range(2)]
        for k, kid in enumerate(kids):
            try:
                _, err = kid.communicate(timeout=120)
            except subprocess.TimeoutExpired:
                for x in kids:
                    x.kill()
                pytest.fail("round %d: process %d still running after 120 s beside another pack on the same device" % (rnd, k))
            assert kid.returncode == 0, (rnd, k, err[-400:])
        times.append(time.monotonic() - t0)
        for k in range(2):
            assert open(os.path.join(t, "o%d.cdata" % k), "rb").read() == want[".cdata"], (rnd, k)
    print("two processes per round, seconds per round:", " ".join("%.1f" % x for x in times))
