"""CPU tests of the host logic and of the C-ABI surface.  No compute is launched on a device here.
The archive-parity checks use the TEST-ONLY host emulation of the kernels (tests/emu, built into
build/libfastore_emu.so) -- the same ppmd_core.h / rc_core.h sources compiled for one lane -- so the
whole host pipeline (reader, unpacker, read-cluster front end, block layout, archive writer) is pinned
against the reference's golden archives without a GPU.  The GPU suite repeats them on the device."""
import ctypes
import os
import re
import struct
import subprocess
import sys

import pytest

from conftest import check_compress_bins_seam, GOLDEN, MODELS, REF_DRIVER, REF_DRIVER_GCC, ROOT, flag_variants, knobs_from_flags, manifest, ref_pipeline

CSRC = os.path.join(ROOT, "fastore_amd", "csrc")


@pytest.fixture(scope="session")
def product_lib():
    subprocess.check_call(["make", "-C", CSRC, "-j", "8"], stdout=subprocess.DEVNULL)
    return ctypes.CDLL(os.path.join(ROOT, "fastore_amd", "libfastore_amd.so"))


@pytest.fixture(scope="session")
def emu_lib():
    subprocess.check_call(["make", "-C", CSRC, "-j", "8", "emu"], stdout=subprocess.DEVNULL)
    import fastore_amd
    return fastore_amd.load_library(os.path.join(ROOT, "build", "libfastore_emu.so"))


def test_c_abi_exports_every_declared_symbol(product_lib):
    hdr = open(os.path.join(ROOT, "include", "fastore_amd.h")).read()
    names = sorted(set(re.findall(r"\b(fsgpu_[a-z_]+)\s*\(", hdr)))
    assert len(names) >= 12
    for n in names:
        assert hasattr(product_lib, n), n


def test_structs_match_header_layout(product_lib):
    import fastore_amd
    assert ctypes.sizeof(fastore_amd.Config) == 96
    assert ctypes.sizeof(fastore_amd.Stats) == 360
    cfg = fastore_amd.Config()
    product_lib.fsgpu_config_defaults(ctypes.byref(cfg))
    # reference defaults: fastore_pack/Params.h:18-147, fastore_bin/Globals.h:61-62
    assert (cfg.min_bin_size, cfg.shift_cost, cfg.mismatch_cost, cfg.max_lz_window, cfg.max_pair_lz_window) == (256, 1, 2, 255, 4096)
    assert (cfg.max_new_variants_per_read, cfg.max_hamming_distance, cfg.min_consensus_size, cfg.world_size) == (1, 8, 10, 1)


def test_no_device_means_loud_failure_not_fallback(product_lib):
    product_lib.fsgpu_device_count.restype = ctypes.c_int
    if product_lib.fsgpu_device_count() > 0:
        pytest.skip("a HIP device is present")
    import fastore_amd
    with pytest.raises(fastore_amd.FastoreError, match="no HIP device"):
        fastore_amd.Packer()


def test_cli_usage_and_errors():
    subprocess.check_call(["make", "-C", CSRC, "-j", "8"], stdout=subprocess.DEVNULL)
    cli = os.path.join(ROOT, "fastore_amd", "fastore_pack")
    r = subprocess.run([cli], capture_output=True)
    assert r.returncode == 255 and b"usage" in r.stderr          # < 3 args: usage, -1 (fastore_pack/main.cpp:26-33)
    r = subprocess.run([cli, "x", "-ia", "-ob"], capture_output=True)
    assert r.returncode == 255
    r = subprocess.run([cli, "e", "-ofoo", "-t1"], capture_output=True)
    assert r.returncode == 255 and b"Error: no input file specified" in r.stderr
    r = subprocess.run([cli, "e", "-ifoo", "-t1"], capture_output=True)
    assert r.returncode == 255 and b"Error: no output file(s) specified" in r.stderr
    r = subprocess.run([cli, "e", "-ifoo", "-obar", "-t99"], capture_output=True)
    assert r.returncode == 255 and b"invalid number of threads" in r.stderr


def test_introsort_equals_libstdcxx_sort(tmp_path):
    exe = str(tmp_path / "t_introsort")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "emu", "test_introsort.cpp")])
    assert subprocess.run([exe], capture_output=True).stdout.strip() == b"OK"


def test_split_pack_worker_slots_keep_caps_and_serve_by_rank(tmp_path):
    # packer.h: HostGate -- the worker slots the pipelines of a split pack share (capi.cpp: packSplit)
    exe = str(tmp_path / "t_hostgate")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "emu", "test_hostgate.cpp")])
    r = subprocess.run([exe], capture_output=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == b"ok", r.stderr


def read_cmeta(path):
    m = open(path, "rb").read()
    foff, fsize = struct.unpack_from("<QQ", m, 0)
    n, = struct.unpack_from("<I", m, foff)
    sizes = struct.unpack_from("<%dQ" % n, m, foff + 4)
    sigs = struct.unpack_from("<%dI" % n, m, foff + 4 + 8 * n)
    conf = m[foff + 4 + 12 * n: foff + 4 + 12 * n + 56]
    tail = m[foff + 4 + 12 * n + 56: foff + fsize]
    # ArchiveConfig fields that are a function of the input (padding / char* members are stack garbage in the reference)
    fields = (conf[0:3], conf[3:11], conf[16:18], conf[24:28], conf[48:56])
    return sizes, sigs, fields, tail, (foff, fsize, len(m))


def assert_same_archive(got_prefix, want_prefix):
    a = open(got_prefix + ".cdata", "rb").read(); b = open(want_prefix + ".cdata", "rb").read()
    assert a == b, ".cdata differs"
    ga, gb = read_cmeta(got_prefix + ".cmeta"), read_cmeta(want_prefix + ".cmeta")
    assert ga == gb, ".cmeta differs field-wise"
    assert sum(ga[0]) == len(a)


@pytest.mark.parametrize("name,paired,flags", manifest())
def test_host_pipeline_reproduces_reference_archives(emu_lib, tmp_path, name, paired, flags):
    import fastore_amd
    with fastore_amd.Packer(lib=emu_lib, host_threads=4, **knobs_from_flags(flags)) as p:
        assert p.device_name == "host-emulation"
        st = p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / "o"))
    assert_same_archive(str(tmp_path / "o"), os.path.join(GOLDEN, name + ".ref"))
    assert st["bins"] > 20 and st["block0_records"] > 0      # both the LZ path and block 0 are exercised


def test_cli_without_a_device_fails_loudly_and_leaves_no_archive(tmp_path):
    # the CLI's context starts its device on a thread of its own (the front end of the first bins does not wait for it): a
    # machine without a GPU must still end in the reference's error convention -- and not in a half-written archive
    import fastore_amd
    if not os.path.exists(fastore_amd.PACK_CLI):
        pytest.skip("CLI not built")
    out = str(tmp_path / "o")
    r = subprocess.run([fastore_amd.PACK_CLI, "e", "-i" + os.path.join(GOLDEN, "se_lossless.in"), "-o" + out, "-r", "-f24", "-c10", "-d8", "-w1024", "-W1024"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode == 0:
        pytest.skip("a HIP device is present")
    assert r.returncode == 255 and b"Error: no HIP device available" in r.stderr
    assert not os.path.exists(out + ".cdata") and not os.path.exists(out + ".cmeta")


@pytest.mark.parametrize("ahead", [65536, 200000, 1 << 20])
def test_archive_extent_reserved_ahead_too_short_or_too_long(emu_lib, tmp_path, monkeypatch, ahead):
    # the archive's pages are reserved from an ESTIMATE while the device works: blocks that end inside the extent are copied
    # into it, the others are written behind it, and the file is cut to its real length -- whatever the estimate was
    import fastore_amd
    monkeypatch.setenv("FS_AHEAD_BYTES", str(ahead))
    name, paired, flags = manifest()[0]
    with fastore_amd.Packer(lib=emu_lib, host_threads=4, **knobs_from_flags(flags)) as p:
        p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / "o"))
    assert_same_archive(str(tmp_path / "o"), os.path.join(GOLDEN, name + ".ref"))


@pytest.mark.parametrize("slices,lanes", [(4, 2), (7, 3), (2, 1)])
def test_sliced_pipeline_does_not_change_the_archive(emu_lib, tmp_path, slices, lanes):
    # a batch cut into slices (front end of slice k+1 overlapping the device work of slice k) yields the same blocks
    import fastore_amd
    for name, paired, flags in manifest()[:2]:
        with fastore_amd.Packer(lib=emu_lib, host_threads=4, pipeline_slices=slices, pipeline_lanes=lanes, **knobs_from_flags(flags)) as p:
            p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / "o"))
        assert_same_archive(str(tmp_path / "o"), os.path.join(GOLDEN, name + ".ref"))


def test_host_build_of_the_qvz_core_matches_reference_vectors(emu_lib):
    import sys
    sys.path.insert(0, GOLDEN)
    import fastore_amd, qvz_inputs
    with fastore_amd.Packer(lib=emu_lib) as p:
        for name in qvz_inputs.CASES:
            lens, quals = qvz_inputs.reads_case(name)
            got = p.qvz_encode(qvz_inputs.footer_for(name), [(lens, quals)])[0]
            assert got == open(os.path.join(GOLDEN, "vectors", name + ".out"), "rb").read(), name
        with pytest.raises(fastore_amd.FastoreError):
            p.qvz_encode(qvz_inputs.qvz_footer()[:500], [(lens, quals)])


def test_missing_input_is_an_error(emu_lib, tmp_path):
    import fastore_amd
    with fastore_amd.Packer(lib=emu_lib) as p:
        with pytest.raises(fastore_amd.FastoreError, match="Cannot open file"):
            p.pack_file(str(tmp_path / "nope"), str(tmp_path / "o"))


def test_truncated_meta_is_an_error(emu_lib, tmp_path):
    import fastore_amd, shutil
    for e in ("bmeta", "bdna", "bqua", "bhead"):
        shutil.copy(os.path.join(GOLDEN, "se_lossless.in." + e), str(tmp_path / ("x." + e)))
    data = open(str(tmp_path / "x.bmeta"), "rb").read()
    open(str(tmp_path / "x.bmeta"), "wb").write(data[:len(data) // 2])
    with fastore_amd.Packer(lib=emu_lib) as p:
        with pytest.raises(fastore_amd.FastoreError):
            p.pack_file(str(tmp_path / "x"), str(tmp_path / "o"))


def test_variable_length_library_is_refused_not_read_out_of_bounds(emu_lib, tmp_path):
    # Reads of different lengths in one cluster: the reference indexes its consensus buffers by the first read's length
    # (ContigBuilder::AddRecord; "TODO: verify for variable-length reads", FastqCompressor.cpp:1766), reads outside them
    # and its archive does not decode back to the input -- there is no defined result to reproduce.  The fixture is the
    # reference's bin-stage output for such a library (tests/golden/make_golden.sh); the product must fail cleanly.
    import fastore_amd
    with fastore_amd.Packer(lib=emu_lib, host_threads=2, min_bin_size=24, max_lz_window=256, max_pair_lz_window=256) as p:
        with pytest.raises(fastore_amd.FastoreError, match="different lengths"):
            p.pack_file(os.path.join(GOLDEN, "se_varlen.in"), str(tmp_path / "o"))


@pytest.mark.parametrize("name,paired,sha,flags", flag_variants())
def test_host_pipeline_under_non_default_flags(emu_lib, tmp_path, name, paired, sha, flags):
    # matcher / consensus knobs away from the C1 profile (-l, -e/-E, -m, -q, -n, tiny windows): same bytes as the reference
    import fastore_amd, hashlib
    with fastore_amd.Packer(lib=emu_lib, host_threads=3, **knobs_from_flags(flags)) as p:
        p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / "o"))
    assert hashlib.sha256(open(str(tmp_path / "o.cdata"), "rb").read()).hexdigest() == sha


@pytest.mark.skipif(not (os.path.exists(REF_DRIVER) and os.path.exists(REF_DRIVER_GCC)), reason="reference binaries (oracle/_ref) not built")
@pytest.mark.parametrize("length,paired,q", [(36, False, 0), (36, True, 2), (250, False, 0), (250, True, 0), (200, False, 3)])
def test_host_pipeline_equals_live_reference_across_read_lengths(emu_lib, tmp_path, length, paired, q):
    # short and long reads (>= 128 bp switches the contig shift coding to absolute positions, FastqCompressor.cpp
    # CompressContigRead), binned and packed by the real reference here, then packed by the host pipeline
    import fastore_amd
    t = str(tmp_path)
    binned, pe = ref_pipeline(t, "lib", 3000, length, length * 60, 300 + length, paired, q, threads=1)
    flags = ["-r", "-f24", "-c10", "-d8", "-w1024", "-W1024"]
    subprocess.check_call([REF_DRIVER, "pack", "-i" + binned, "-o" + os.path.join(t, "ref"), "-t1"] + flags + pe)
    with fastore_amd.Packer(lib=emu_lib, host_threads=3, **knobs_from_flags(flags)) as p:
        st = p.pack_file(binned, os.path.join(t, "emu"))
    assert open(os.path.join(t, "emu.cdata"), "rb").read() == open(os.path.join(t, "ref.cdata"), "rb").read()
    assert st["bins"] > 5


@pytest.mark.parametrize("name,paired,flags", manifest())
def test_compress_bins_seam_on_the_host_pipeline(emu_lib, name, paired, flags):
    # the per-bin C-ABI seam (what FastqCompressor::Compress would bind): flat batch in, one block per bin out
    import fastore_amd
    with fastore_amd.Packer(lib=emu_lib, host_threads=3, **knobs_from_flags(flags)) as p:
        check_compress_bins_seam(fastore_amd, p, name, flags, lib=emu_lib)


def test_host_build_of_the_rc_core_rejects_symbols_outside_the_alphabet(emu_lib):
    # same core header as the device: a symbol without a statistic must end the stream, not loop in the normalisation
    import fastore_amd
    with fastore_amd.Packer(lib=emu_lib) as p:
        assert len(p.rc_encode([MODELS["a8o4"][0]], [bytes([1, 0, 3, 1, 7, 2])])[0]) >= 8
        for name, bad in (("s2o4", bytes([2, 0])), ("a8o4", bytes([1, 0, 9, 0])), ("a8o6", bytes([0, 0, 200, 1])), ("a2o10", bytes([1, 0, 0, 3]))):
            with pytest.raises(fastore_amd.FastoreError, match="alphabet"):
                p.rc_encode([MODELS[name][0]], [bad])
        with pytest.raises(fastore_amd.FastoreError, match="alphabet"):
            p.rc_encode([6], [bytes([65, 3, 66, 64])])


@pytest.mark.timeout(120)
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_damaged_inputs_end_in_an_error_or_an_archive_never_a_hang(emu_lib, tmp_path, seed):
    # bit flips, truncation, runs of 0x00 / 0xFF in each of the four stream files (a longer run of the same generator
    # went through 1 000 cases under AddressSanitizer).  Whatever comes out, the call has to return: the footer totals
    # are held against the file sizes, group sizes against the bin, and a range-coder symbol outside its alphabet stops
    # its stream instead of spinning in the coder (on the device that would be a wave that never ends).
    import fastore_amd, random, shutil
    rnd = random.Random(seed)
    with fastore_amd.Packer(lib=emu_lib, host_threads=2, min_bin_size=24) as p:
        for it in range(12):
            name = rnd.choice(["se_lossless", "pe_lossless", "se_qvz", "se_reduced", "se_noheader"])
            for e in ("bmeta", "bdna", "bqua", "bhead"):
                src = os.path.join(GOLDEN, name + ".in." + e)
                dst = str(tmp_path / ("x." + e))
                if os.path.exists(src):
                    shutil.copy(src, dst)
                elif os.path.exists(dst):
                    os.remove(dst)
            target = rnd.choice(["bmeta", "bmeta", "bmeta", "bdna", "bqua", "bhead"])
            path = str(tmp_path / ("x." + target))
            if not os.path.exists(path):
                continue
            d = bytearray(open(path, "rb").read())
            mode = rnd.choice(["flip", "flip", "trunc", "zero", "ff"])
            if mode == "flip":
                for _ in range(rnd.randint(1, 8)):
                    i = rnd.randrange(len(d)); d[i] ^= 1 << rnd.randrange(8)
            elif mode == "trunc":
                d = d[:rnd.randrange(len(d))]
            else:
                i = rnd.randrange(len(d)); n = rnd.randint(1, 64); d[i:i + n] = bytes([0 if mode == "zero" else 255]) * n
            open(path, "wb").write(d)
            try:
                p.pack_file(str(tmp_path / "x"), str(tmp_path / "o"))
            except fastore_amd.FastoreError:
                pass
        # the context is still good after the failures
        p.pack_file(os.path.join(GOLDEN, "se_lossless.in"), str(tmp_path / "good"))
        assert_same_archive(str(tmp_path / "good"), os.path.join(GOLDEN, "se_lossless.ref"))


def test_compress_bins_rejects_a_malformed_batch(emu_lib):
    # the seam takes caller-built tables: indices and positions are checked before the front end walks them
    import ctypes as C, struct
    import fastore_amd
    with fastore_amd.Packer(lib=emu_lib, host_threads=2, min_bin_size=24) as p, \
            fastore_amd.Library(os.path.join(GOLDEN, "se_lossless.in"), 24, lib=emu_lib) as L:
        p.set_archive_params(L.config, L.header_fields, L.quality_codebook)
        good = L.batch
        def variant(**kw):
            b = fastore_amd.BinBatch()
            C.memmove(C.byref(b), C.byref(good), C.sizeof(b))
            keep = []
            for field, (index, fmt, offset, value, size) in kw.items():
                n = getattr(b, "n_" + field)
                raw = bytearray(C.string_at(getattr(b, field), n * size))
                struct.pack_into(fmt, raw, index * size + offset, value)
                buf = (C.c_char * len(raw)).from_buffer(raw); keep.append((raw, buf))
                setattr(b, field, C.addressof(buf))
            return b, keep
        cases = {
            "signature outside the read": dict(records=(3, "<H", 12, 250, 16)),          # minim_pos of record 3
            "record outside the batch": dict(records=(0, "<I", 0, 0xFFFFFF00, 16)),        # seq_off
            "node references outside": dict(nodes=(1, "<I", 0, 0x7FFFFFFF, 20)),           # rec of node 1
            "bin ranges outside": dict(bins=(0, "<I", 28, 0x7FFFFFFF, 40)),                # rec_count of bin 0
        }
        for msg, kw in cases.items():
            b, keep = variant(**kw)
            with pytest.raises(fastore_amd.FastoreError, match=msg):
                p.compress_bins(b)
        assert len(p.compress_bins(good)) == good.n_bins                                   # and the context still works


@pytest.mark.parametrize("batch_bases", [40_000, 300_000])
def test_libraries_larger_than_a_device_batch(emu_lib, tmp_path, batch_bases):
    # a library that does not fit one device batch is packed in several (the 100 M-pair configurations); blocks of
    # earlier batches are routed to their archives while later ones are coded -- same bytes, for one library and for
    # several libraries sharing the batches
    import fastore_amd
    fx = manifest()[:3]
    with fastore_amd.Packer(lib=emu_lib, host_threads=3, batch_bases=batch_bases, **knobs_from_flags(fx[0][2])) as p:
        for name, paired, flags in fx:
            st = p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / name))
            assert_same_archive(str(tmp_path / name), os.path.join(GOLDEN, name + ".ref"))
        ins = [os.path.join(GOLDEN, name + ".in") for name, _, _ in fx]
        outs = [str(tmp_path / ("m_" + name)) for name, _, _ in fx]
        p.pack_files(ins, outs)
        for (name, _, _), o in zip(fx, outs):
            assert_same_archive(o, os.path.join(GOLDEN, name + ".ref"))


@pytest.mark.parametrize("gpus", [2, 3])
def test_cli_packs_one_library_on_several_devices(tmp_path, gpus):
    # fastore_pack e ... -G<n>: one context per device packs its share of the bins, the parts are merged in -t1 order
    cli = os.path.join(ROOT, "build", "fastore_pack_emu")
    for name, paired, flags in manifest()[:2]:
        out = str(tmp_path / ("g_" + name))
        r = subprocess.run([cli, "e", "-i" + os.path.join(GOLDEN, name + ".in"), "-o" + out, "-t4", "-G%d" % gpus] + flags + (["-z"] if paired else []), capture_output=True)
        assert r.returncode == 0, r.stderr
        assert_same_archive(out, os.path.join(GOLDEN, name + ".ref"))
        assert not [f for f in os.listdir(str(tmp_path)) if ".part" in f]


@pytest.mark.parametrize("world", [2, 5])
def test_shard_api_lpt_split_and_positional_writes(emu_lib, tmp_path, world):
    # fsgpu_shard_pack / _table / _write: every rank holds its LPT share, the size tables are summed (the one collective
    # of the multi-GPU path), the ranks write their blocks in ANY order -- the archive is the single-writer archive
    import fastore_amd, numpy as np
    name, paired, flags = manifest()[0]
    out = str(tmp_path / "o")
    open(out + ".cdata", "wb").write(b"x" * (8 << 20))            # a longer file of an earlier run must not survive
    packers = [fastore_amd.Packer(lib=emu_lib, host_threads=2, rank=r, world_size=world, **knobs_from_flags(flags)) for r in range(world)]
    tables = [p.shard_pack(os.path.join(GOLDEN, name + ".in")) for p in packers]
    sigs = tables[0][0]
    assert all((t[0] == sigs).all() for t in tables)
    assert sigs[0] == sigs.max() and (np.diff(sigs[1:].astype(np.int64)) > 0).all()         # block 0, then ascending signature
    own = np.stack([t[1] for t in tables])
    assert ((own > 0).sum(axis=0) == 1).all()                                               # every block has exactly one owner
    records = [p.stats()["records"] for p in packers]
    assert max(records) - min(records) <= 0.15 * sum(records) / world + 4000, records      # LPT: the shares are even
    total = own.sum(axis=0)
    for r in reversed(range(world)):                                                        # last rank first
        packers[r].shard_write(out, total)
    for p in packers:
        p.close()
    assert_same_archive(out, os.path.join(GOLDEN, name + ".ref"))


def test_cli_several_devices_verbose_statistics(tmp_path):
    # -G<n> -v: the StreamSizes statistics are those of the whole archive (taken from the merged blocks' headers)
    cli = os.path.join(ROOT, "build", "fastore_pack_emu")
    name, paired, flags = manifest()[0]
    r = subprocess.run([cli, "e", "-i" + os.path.join(GOLDEN, name + ".in"), "-o" + str(tmp_path / "o"), "-t4", "-G3", "-v"] + flags, capture_output=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == open(os.path.join(GOLDEN, name + ".ref.vout"), "rb").read()
    assert b"Parts processed" in r.stderr


def test_merge_parts_of_rank_sharded_contexts(emu_lib, tmp_path):
    import fastore_amd
    name, paired, flags = manifest()[1]
    out = str(tmp_path / "o")
    for r in range(3):
        with fastore_amd.Packer(lib=emu_lib, host_threads=2, rank=r, world_size=3, **knobs_from_flags(flags)) as p:
            p.pack_file(os.path.join(GOLDEN, name + ".in"), out)
    fastore_amd.merge_parts(out, 3, lib=emu_lib)
    assert_same_archive(out, os.path.join(GOLDEN, name + ".ref"))
    with pytest.raises(fastore_amd.FastoreError, match="Cannot open"):
        fastore_amd.merge_parts(out, 3, lib=emu_lib)            # the parts are gone


@pytest.mark.parametrize("name,paired", [("se_lossless", False), ("pe_lossless", True)])
def test_cli_verbose_statistics_match_the_reference(tmp_path, name, paired):
    # -v: "StreamSizes:" + compressed bytes per stream summed over the standard blocks + the raw block's four sizes
    cli = os.path.join(ROOT, "build", "fastore_pack_emu")
    flags = [f for n, p, f in manifest() if n == name][0]
    r = subprocess.run([cli, "e", "-i" + os.path.join(GOLDEN, name + ".in"), "-o" + str(tmp_path / "o"), "-t3", "-v"] + flags + (["-z"] if paired else []), capture_output=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == open(os.path.join(GOLDEN, name + ".ref.vout"), "rb").read()
    assert b"Parts processed" in r.stderr


def test_quality_gather_emulation_matches_the_restated_unpack(emu_lib):
    # the test-only host form of fs_gather_quality (tests/emu/engine_emu.cpp) against the numpy restatement: pins the
    # expectation the GPU test uses, and the descriptor plumbing of fsgpu_gather_quality
    import fastore_amd
    from conftest import quality_gather_case
    with fastore_amd.Packer(lib=emu_lib, device_id=0) as p:
        for seed in (1, 2, 3):
            packed, strings, expect = quality_gather_case(seed)
            assert p.gather_quality(packed, strings) == expect
        assert p.gather_quality(b"", []) == b""
        with pytest.raises(fastore_amd.FastoreError, match="outside the packed scores"):
            p.gather_quality(b"\0" * 8, [(40, 5, False)])


def test_binned_quality_gather_emulation_matches_the_restated_symbolisation(emu_lib):
    import fastore_amd
    from conftest import quality_gather_binned_case
    with fastore_amd.Packer(lib=emu_lib, device_id=0) as p:
        for bits, thr in ((3, 20), (1, 20), (1, 5), (1, 41)):
            packed, strings, expect = quality_gather_binned_case(bits + thr, bits, thr)
            assert p.gather_quality_binned(packed, bits, thr, strings) == expect, (bits, thr)


@pytest.mark.parametrize("name,paired,flags", [m for m in manifest() if m[0] in ("se_lossless", "pe_lossless", "se_noheader", "se_c0", "se_reduced", "se_binary")])
def test_device_quality_path_is_taken_and_changes_no_byte(emu_lib, tmp_path, monkeypatch, name, paired, flags):
    # lossless archives packed from .b* files keep their scores packed: the quality streams are gathered by the engine
    # (fs_gather_quality on the device), not symbolised by the host front end -- same archive either way
    import fastore_amd
    ref = open(os.path.join(GOLDEN, name + ".ref.cdata"), "rb").read()
    seen = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FS_DEVICE_QUALITY", mode)
        with fastore_amd.Packer(lib=emu_lib, device_id=0, **knobs_from_flags(flags)) as p:
            st = p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / ("o" + mode)))
        assert open(str(tmp_path / ("o" + mode)) + ".cdata", "rb").read() == ref
        seen[mode] = st["gather_symbols"]
    assert seen["0"] == 0 and seen["1"] > 0


@pytest.mark.parametrize("name,paired,flags", manifest())
def test_matcher_table_and_trace_against_the_scalar_restatement(emu_lib, name, paired, flags):
    # the table of match-tree constructions the front end hands to the device matcher, and the rows the host scan traces
    # for the parity check, against the test-only scalar window search (tests/emu/engine_emu.cpp): no GPU needed to pin
    # the window model (last W-1 non-duplicate reads, root copy, dummy slots) that matcher.hip implements
    import fastore_amd
    for window in (None, 3, 17):
        kn = knobs_from_flags(flags)
        if window:
            kn["max_lz_window"] = window
        with fastore_amd.Packer(lib=emu_lib, device_id=0, **kn) as p:
            reads, differing = p.matcher_check(os.path.join(GOLDEN, name + ".in"))
        assert reads > 1000 and differing == 0, (window, reads, differing)


@pytest.mark.parametrize("name,paired,flags", manifest())
def test_packed_bases_and_their_descriptors_against_the_unpacked_ones(emu_lib, tmp_path, monkeypatch, name, paired, flags):
    # the bin's .bdna bytes + one descriptor per read (bit offset, two- or three-bit form, where the signature stands and which)
    # that the window search gets instead of ASCII bases: the test-only stand-in reads them base by base, as fs_unpack_planes
    # does, and holds every base against the host's unpacked one; the pack takes that way by default and changes no byte
    import fastore_amd
    with fastore_amd.Packer(lib=emu_lib, device_id=0, **knobs_from_flags(flags)) as p:
        bases, differing, reads, rows = p.unpack_check(os.path.join(GOLDEN, name + ".in"))
    assert bases > 100_000 and differing == 0 and reads > 1000 and rows == 0, (bases, differing, reads, rows)
    ref = open(os.path.join(GOLDEN, name + ".ref.cdata"), "rb").read()
    seen = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FS_DEVICE_UNPACK", mode)
        with fastore_amd.Packer(lib=emu_lib, device_id=0, **knobs_from_flags(flags)) as p:
            st = p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / ("o" + mode)))
        assert open(str(tmp_path / ("o" + mode)) + ".cdata", "rb").read() == ref
        seen[mode] = st
    assert seen["0"]["matcher_unpacked_reads"] == 0 and seen["0"]["matcher_reads"] > 0
    assert seen["1"]["matcher_unpacked_reads"] == seen["1"]["matcher_reads"] > 0
    assert seen["1"]["matcher_bases_h2d_bytes"] < 0.45 * seen["0"]["matcher_bases_h2d_bytes"]


@pytest.mark.parametrize("name,paired,flags", manifest()[:3])
def test_library_of_several_batches_goes_through_two_pipelines(emu_lib, tmp_path, monkeypatch, name, paired, flags):
    # more standard-bin bases than one device batch holds: fsgpu_pack_file deals the heaviest bins (one batch) to the context and
    # all the others to a helper context, both pack at once and write their blocks at their places -- same archive as one
    # pipeline gives, and as the reference's
    import fastore_amd
    ref = open(os.path.join(GOLDEN, name + ".ref.cdata"), "rb").read()
    for split in ("1", "0"):
        monkeypatch.setenv("FS_SPLIT_PIPELINES", split)
        with fastore_amd.Packer(lib=emu_lib, device_id=0, batch_bases=300_000, **knobs_from_flags(flags)) as p:
            st = p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / ("s" + split)))
            assert st["device_batches"] >= 2, st["device_batches"]
            again = p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / ("t" + split)))       # the helper context is reused
        for o in ("s", "t"):
            assert open(str(tmp_path / (o + split)) + ".cdata", "rb").read() == ref, (split, o)
        assert again["bins"] == 2 * st["bins"]


@pytest.mark.parametrize("name,paired,flags", [m for m in manifest() if m[1]])
def test_mate_search_rows_against_the_scalar_restatement(emu_lib, name, paired, flags):
    # paired-end bins: the pairs the front end hands to the device mate search (fs_match_mates) in walk order, and the rows the
    # host's search traces for the parity check, against the test-only scalar history search (tests/emu/engine_emu.cpp) -- the
    # model the kernel implements (minimizer sets of the mate's halves, entries oldest first, four stored positions each,
    # identical mates to the back), pinned without a GPU; small histories make the ring wrap many times
    import fastore_amd
    for window in (None, 2, 5, 64):
        kn = knobs_from_flags(flags)
        if window:
            kn["max_pair_lz_window"] = window
        with fastore_amd.Packer(lib=emu_lib, device_id=0, **kn) as p:
            pairs, differing = p.pe_matcher_check(os.path.join(GOLDEN, name + ".in"))
        assert pairs > 1000 and differing == 0, (window, pairs, differing)


@pytest.mark.parametrize("name,paired,flags", [m for m in manifest() if m[1]])
def test_device_mate_search_is_taken_and_changes_no_byte(emu_lib, tmp_path, monkeypatch, name, paired, flags):
    # the archive with the mate searches taken from the "device" (here: the emulation's stand-in) -- handed over and searched in batches of
    # bins while the host threads go on (2: fs::PendingPairs, the mate streams written from the rows when a batch comes back; several
    # slices and few threads, so that bins complete in every order) -- and with the host's search
    import fastore_amd
    ref = open(os.path.join(GOLDEN, name + ".ref.cdata"), "rb").read()
    for mode, kw in (("2", {}), ("2", dict(host_threads=3, pipeline_slices=5, pipeline_lanes=2)), ("0", {})):
        monkeypatch.setenv("FS_DEVICE_MATES", mode)
        with fastore_amd.Packer(lib=emu_lib, device_id=0, **kw, **knobs_from_flags(flags)) as p:
            st = p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / ("o" + mode)))
        assert open(str(tmp_path / ("o" + mode)) + ".cdata", "rb").read() == ref, (mode, kw)
        assert (st["mate_pairs"] > 1000) == (mode != "0"), (mode, st["mate_pairs"])      # ... and they were taken there, every pair of a standard bin


@pytest.mark.parametrize("name,paired,flags", [m for m in manifest() if m[0] != "se_noheader"])
def test_device_tokeniser_is_taken_and_changes_no_byte(emu_lib, tmp_path, monkeypatch, name, paired, flags):
    # archives with read ids packed from .b* files keep their headers packed: the IdToken / IdValue streams are written by the
    # engine (fs_tokenise_ids on the device), not by the host front end -- same archive either way
    import fastore_amd
    ref = open(os.path.join(GOLDEN, name + ".ref.cdata"), "rb").read()
    seen = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FS_DEVICE_IDS", mode)
        with fastore_amd.Packer(lib=emu_lib, device_id=0, **knobs_from_flags(flags)) as p:
            st = p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / ("o" + mode)))
        assert open(str(tmp_path / ("o" + mode)) + ".cdata", "rb").read() == ref
        seen[mode] = st["tokenised_ids"]
    has_ids = open(os.path.join(GOLDEN, name + ".in.bmeta"), "rb").read(33)[32] != 0       # usesHeaderStream of the .bmeta header
    assert seen["0"] == 0 and seen["1"] == (st["records"] if has_ids else 0), seen


def test_tokeniser_checker_against_the_scalar_restatement(emu_lib):
    # the packed-header descriptors, the field-table blob and the parity harness of fs_tokenise_ids, with the test-only scalar
    # stand-in of the kernel: every standard bin of the golden libraries that carry read ids
    import fastore_amd
    total = 0
    for name, paired, flags in manifest():
        with fastore_amd.Packer(lib=emu_lib, device_id=0, **knobs_from_flags(flags)) as p:
            ids, differing = p.tokeniser_check(os.path.join(GOLDEN, name + ".in"))
        assert differing == 0, (name, ids, differing)
        total += ids
    assert total > 10000


REF_CLI = os.path.join(ROOT, "oracle", "_ref", "fastore_pack_ref")


@pytest.mark.skipif(not os.path.exists(REF_CLI), reason="the reference's own CLI (oracle/_ref/fastore_pack_ref) not built")
def test_cli_matches_the_reference_cli(tmp_path):
    # the reference's OWN command line (its main.cpp, compiled unmodified by oracle/Makefile) beside the product's: the same exit
    # status and the same `Error: ...` line for every malformed call, usage on stderr for too few arguments -- and, on a golden
    # library, the same archive and the same -v statistics on stdout (the product's CLI here over the test-only emulation:
    # the device build's is held against the same files in the GPU suite)
    subprocess.check_call(["make", "-C", CSRC, "-j", "8", "emu"], stdout=subprocess.DEVNULL)
    ours = os.path.join(ROOT, "build", "fastore_pack_emu")
    cases = [[], ["e"], ["x", "-ia", "-ob"], ["e", "-ofoo", "-t1"], ["e", "-ifoo", "-t1"], ["e", "-ifoo", "-obar", "-t99"], ["e", "-ifoo", "-obar", "-t0"],
             ["e", "-i" + str(tmp_path / "nothing_here"), "-o" + str(tmp_path / "o"), "-t1"],
             ["e", "-i" + os.path.join(GOLDEN, manifest()[0][0] + ".in"), "-o" + str(tmp_path / "no_such_dir" / "o"), "-t1"] + manifest()[0][2]]
    for argv in cases:
        a = subprocess.run([ours] + argv, capture_output=True); b = subprocess.run([REF_CLI] + argv, capture_output=True)
        assert a.returncode == b.returncode == 255, (argv, a.returncode, b.returncode)
        err = lambda r: [l for l in r.stderr.decode(errors="replace").splitlines() if l.startswith("Error:")]
        assert err(a) == err(b), (argv, err(a), err(b))
        assert (b"usage" in a.stderr) == (b"usage" in b.stderr), argv
    name, paired, flags = manifest()[0]
    pe = ["-z"] if paired else []
    a = subprocess.run([ours, "e", "-i" + os.path.join(GOLDEN, name + ".in"), "-o" + str(tmp_path / "a"), "-t1", "-v"] + flags + pe, capture_output=True)
    b = subprocess.run([REF_CLI, "e", "-i" + os.path.join(GOLDEN, name + ".in"), "-o" + str(tmp_path / "b"), "-t1", "-v"] + flags + pe, capture_output=True, timeout=300)
    assert a.returncode == b.returncode == 0, (a.stderr[-300:], b.stderr[-300:])
    assert a.stdout == b.stdout                                    # StreamSizes: ... (CompressorModule.cpp:357-441)
    assert open(str(tmp_path / "a.cdata"), "rb").read() == open(str(tmp_path / "b.cdata"), "rb").read()
    assert_same_archive(str(tmp_path / "a"), str(tmp_path / "b"))


@pytest.mark.parametrize("name,paired,flags", manifest())
def test_device_emission_ops_and_the_host_walk_write_the_same_streams(emu_lib, tmp_path, monkeypatch, name, paired, flags):
    # device-side emission (fsdev::EmitOp, emit_core.h): the walk leaves an op where it would compare bases, and the streams that
    # hold bases -- HardReads, LettersX, Match, MatchBinary, CMatch, CLetters, the paired-end ones, the run-length coded LZ ids -- are
    # written from the ops.  Here through the emulation's serial form of emit_core.h (the kernels' own source): pre-entropy bytes
    # of every standard bin against the walk's own, and the archive with the ops and without (FS_DEVICE_EMIT=0)
    import fastore_amd
    with fastore_amd.Packer(lib=emu_lib, device_id=0, **knobs_from_flags(flags)) as p:
        ops, streams, differing = p.emit_check(os.path.join(GOLDEN, name + ".in"))
    assert ops > 1000 and streams >= 7 * 20 and differing == 0, (ops, streams, differing)
    # (the kernels' own form -- a wavefront per op, emit_wave.h -- runs on the lock-step emulation: tests/test_simt.py)
    ref = open(os.path.join(GOLDEN, name + ".ref.cdata"), "rb").read()
    for mode in ("1", "0"):
        monkeypatch.setenv("FS_DEVICE_EMIT", mode)
        with fastore_amd.Packer(lib=emu_lib, device_id=0, **knobs_from_flags(flags)) as p:
            p.pack_file(os.path.join(GOLDEN, name + ".in"), str(tmp_path / ("o" + mode)))
        assert open(str(tmp_path / ("o" + mode)) + ".cdata", "rb").read() == ref, mode


@pytest.mark.parametrize("name,paired,flags", manifest()[:2])
def test_a_slice_that_would_outgrow_its_address_space_falls_back_to_the_walks_own_streams(emu_lib, tmp_path, name, paired, flags):
    # A slice lives in one 32-bit address space on the device; a bin whose base-holding streams the device writes costs ~7 bytes a base
    # there.  Every bin claims its footprint from its slice's budget before its walk starts, and a bin that finds the budget used up keeps
    # the walk's own streams instead of failing the pack ("emitted streams larger than 4 GiB" with few slices and > 0.6 GB of bases).
    # Here with a budget of a few bins' worth and ONE slice: some bins go one way, some the other, the archive is the reference's.
    code = ("import os, sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import fastore_amd; from conftest import knobs_from_flags\n"
            "lib = fastore_amd.load_library(%r)\n"
            "with fastore_amd.Packer(lib=lib, device_id=0, pipeline_slices=1, **knobs_from_flags(%r)) as p: p.pack_file(%r, %r)" %
            (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "build", "libfastore_emu.so"), list(flags), os.path.join(GOLDEN, name + ".in"), str(tmp_path / "o")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, env=dict(os.environ, FS_SLICE_DEV_CAP=str(2 << 20)))
    assert r.returncode == 0, r.stderr
    assert open(str(tmp_path / "o.cdata"), "rb").read() == open(os.path.join(GOLDEN, name + ".ref.cdata"), "rb").read()
