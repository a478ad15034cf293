"""fastore_amd -- MI355X-native `fastore_pack` hot path (ctypes binding of include/fastore_amd.h).

The product is the C-ABI shared library ``libfastore_amd.so`` (host C++ + HIP kernels for gfx950)
built in-tree by ``fastore_amd/csrc/Makefile``.  This module is plumbing only: it loads that library
and fails loudly when it -- or a HIP device -- is missing.  There is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfastore_amd.so")
PACK_CLI = os.path.join(_HERE, "fastore_pack")


class FastoreError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("min_bin_size", C.c_uint32), ("encode_threshold", C.c_int32), ("pair_encode_threshold", C.c_int32),
                ("shift_cost", C.c_int32), ("mismatch_cost", C.c_int32), ("max_lz_window", C.c_uint32),
                ("max_pair_lz_window", C.c_uint32), ("extra_reduce_hard_reads", C.c_uint32),
                ("extra_reduce_expensive_lz", C.c_uint32), ("max_record_shift_diff", C.c_uint32),
                ("max_new_variants_per_read", C.c_uint32), ("max_hamming_distance", C.c_uint32),
                ("min_consensus_size", C.c_uint32), ("device_id", C.c_int32), ("host_threads", C.c_uint32),
                ("max_waves", C.c_uint32), ("batch_bases", C.c_uint64), ("rank", C.c_uint32), ("world_size", C.c_uint32),
                ("pipeline_slices", C.c_uint32), ("pipeline_lanes", C.c_uint32), ("one_shot", C.c_uint32), ("reserved1", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("encode_kernel_ms", C.c_double), ("assemble_kernel_ms", C.c_double), ("frontend_ms", C.c_double),
                ("block0_ms", C.c_double), ("io_ms", C.c_double), ("total_ms", C.c_double),
                ("kernel_launches", C.c_uint64), ("stream_items", C.c_uint64), ("ppmd_symbols", C.c_uint64),
                ("rc_symbols", C.c_uint64), ("ppmd_restarts", C.c_uint64), ("h2d_bytes", C.c_uint64),
                ("d2h_bytes", C.c_uint64), ("bins", C.c_uint64), ("records", C.c_uint64),
                ("algorithmic_bytes", C.c_uint64), ("block0_records", C.c_uint64), ("block0_bytes", C.c_uint64),
                ("cdata_bytes", C.c_uint64), ("host_coded_symbols", C.c_uint64), ("host_coded_streams", C.c_uint64),
                ("ppmd_window_attempts", C.c_uint64), ("ppmd_windows", C.c_uint64), ("ppmd_window_symbols", C.c_uint64),
                ("ppmd_window_rounds", C.c_uint64), ("ppmd_windows_redone", C.c_uint64), ("ppmd_window_light_rounds", C.c_uint64),
                ("gather_kernel_ms", C.c_double), ("gather_symbols", C.c_uint64), ("gather_bytes", C.c_uint64),
                ("matcher_reads", C.c_uint64), ("matcher_call_ms", C.c_double), ("matcher_kernel_ms", C.c_double), ("tokenised_ids", C.c_uint64),
                ("device_batches", C.c_uint64), ("ppmd_max_restarts", C.c_uint64),
                ("stolen_bins", C.c_uint64),
                ("matcher_bases_h2d_bytes", C.c_uint64), ("matcher_unpacked_reads", C.c_uint64),
                ("mate_pairs", C.c_uint64), ("mate_call_ms", C.c_double), ("mate_kernel_ms", C.c_double),
                ("struct_bytes", C.c_uint64), ("ppmd_window_drops", C.c_uint64), ("coder_tail_launches", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class BinBatch(C.Structure):
    """fsgpu_bin_batch: unpacked reads and stored match graph of a batch of standard bins (include/fastore_amd.h)"""
    _fields_ = [("bases", C.c_void_p), ("quals", C.c_void_p), ("heads", C.c_void_p), ("n_bases", C.c_size_t), ("n_heads", C.c_size_t),
                ("records", C.c_void_p), ("n_records", C.c_size_t), ("nodes", C.c_void_p), ("n_nodes", C.c_size_t),
                ("top_nodes", C.c_void_p), ("n_top_nodes", C.c_size_t), ("em_records", C.c_void_p), ("n_em_records", C.c_size_t),
                ("trees", C.c_void_p), ("n_trees", C.c_size_t), ("bins", C.c_void_p), ("n_bins", C.c_size_t)]


class BlockBatch(C.Structure):
    _fields_ = [("data", C.c_void_p), ("sizes", C.POINTER(C.c_uint64)), ("n_blocks", C.c_size_t)]


_lib = None


os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # one hardware queue per pipeline lane (see engine.hip)


def load_library(path=None):
    """Load libfastore_amd.so (in-tree).  Raises FastoreError when it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise FastoreError("%s not found: build it with `make -C fastore_amd/csrc` (or __graft_entry__.build())" % p)
    lib = C.CDLL(p)
    lib.fsgpu_config_defaults.argtypes = [C.POINTER(Config)]
    lib.fsgpu_device_count.restype = C.c_int
    lib.fsgpu_create.restype = C.c_void_p
    lib.fsgpu_create.argtypes = [C.POINTER(Config)]
    lib.fsgpu_create_error.restype = C.c_char_p
    lib.fsgpu_destroy.argtypes = [C.c_void_p]
    lib.fsgpu_last_error.restype = C.c_char_p
    lib.fsgpu_last_error.argtypes = [C.c_void_p]
    lib.fsgpu_device_name.restype = C.c_char_p
    lib.fsgpu_device_name.argtypes = [C.c_void_p]
    lib.fsgpu_pack_file.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
    lib.fsgpu_pack_files.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    lib.fsgpu_reset_stats.argtypes = [C.c_void_p]
    lib.fsgpu_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    lib.fsgpu_get_window_profile.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 8)]
    lib.fsgpu_get_serial_profile.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 2)]
    lib.fsgpu_set_archive_params.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    pp = C.POINTER(C.c_char_p)
    lib.fsgpu_ppmd_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fsgpu_gather_quality_binned.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.fsgpu_tokeniser_check.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.fsgpu_emit_check.argtypes = [C.c_void_p, C.c_char_p] + [C.POINTER(C.c_uint64)] * 3
    lib.fsgpu_matcher_check.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.fsgpu_pe_matcher_check.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.fsgpu_unpack_check.argtypes = [C.c_void_p, C.c_char_p] + [C.POINTER(C.c_uint64)] * 4
    lib.fsgpu_gather_quality.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.fsgpu_rc_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fsgpu_set_quality_codebook.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.fsgpu_qvz_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    del pp
    lib.fsgpu_library_open.restype = C.c_void_p
    lib.fsgpu_library_open.argtypes = [C.c_char_p, C.c_uint32]
    lib.fsgpu_library_close.argtypes = [C.c_void_p]
    lib.fsgpu_library_std_bins.restype = C.POINTER(BinBatch)
    lib.fsgpu_library_std_bins.argtypes = [C.c_void_p]
    for f in (lib.fsgpu_library_config, lib.fsgpu_library_header_fields, lib.fsgpu_library_quality_codebook):
        f.restype = C.c_void_p
        f.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    lib.fsgpu_compress_bins.argtypes = [C.c_void_p, C.POINTER(BinBatch), C.POINTER(BlockBatch)]
    lib.fsgpu_merge_parts.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_size_t]
    lib.fsgpu_shard_pack.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_size_t)]
    lib.fsgpu_shard_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.fsgpu_shard_write.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
    lib.fsgpu_shard_pack_set.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.fsgpu_shard_table_of.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.fsgpu_shard_write_of.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_void_p, C.c_size_t]
    if path is None:
        _lib = lib
    return lib


# C1 profile of the reference's scripts/fastore_compress.sh:146-148: -r -f256 -c10 -d8 -w1024 -W1024
C1_PROFILE = dict(extra_reduce_hard_reads=1, min_bin_size=256, min_consensus_size=10, max_hamming_distance=8,
                  max_lz_window=1024, max_pair_lz_window=1024)


def merge_parts(out_prefix, world_size, lib=None):
    """Merge <out_prefix>.part<r>.{cdata,cmeta}, r < world_size, written by rank-sharded Packers into one archive."""
    lib = lib or load_library()
    err = C.create_string_buffer(256)
    if lib.fsgpu_merge_parts(out_prefix.encode(), world_size, err, len(err)) != 0:
        raise FastoreError("fsgpu_merge_parts: " + err.value.decode())


class Library:
    """The standard bins of a binned library, unpacked into the flat batch that fsgpu_compress_bins takes, plus the
    archive-level parameters of its .bmeta footer (host only; see fsgpu_library_open in include/fastore_amd.h)."""

    def __init__(self, in_prefix, min_bin_size=256, lib=None):
        self.lib = lib or load_library()
        self.h = self.lib.fsgpu_library_open(in_prefix.encode(), min_bin_size)
        if not self.h:
            raise FastoreError("fsgpu_library_open failed: " + self.lib.fsgpu_create_error().decode())

    def close(self):
        if self.h:
            self.lib.fsgpu_library_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def batch(self):
        return self.lib.fsgpu_library_std_bins(self.h).contents

    def _blob(self, fn):
        n = C.c_size_t(0)
        p = fn(self.h, C.byref(n))
        return C.string_at(p, n.value) if p and n.value else b""

    @property
    def config(self):
        return self._blob(self.lib.fsgpu_library_config)

    @property
    def header_fields(self):
        return self._blob(self.lib.fsgpu_library_header_fields)

    @property
    def quality_codebook(self):
        return self._blob(self.lib.fsgpu_library_quality_codebook)

    def signatures(self):
        b = self.batch
        raw = C.string_at(b.bins, b.n_bins * 40)         # sizeof(fsgpu_bin) = 40
        import struct
        return [struct.unpack_from("<I", raw, 40 * i)[0] for i in range(b.n_bins)]


class Packer:
    """One context = one GPU (mirrors one reference FastqCompressor instance per worker)."""

    def __init__(self, device_id=0, host_threads=0, rank=0, world_size=1, lib=None, **knobs):
        self.lib = lib or load_library()
        cfg = Config()
        self.lib.fsgpu_config_defaults(C.byref(cfg))
        for k, v in dict(C1_PROFILE, **knobs).items():
            if not hasattr(cfg, k):
                raise FastoreError("unknown pack option %r" % k)
            setattr(cfg, k, v)
        cfg.device_id, cfg.host_threads, cfg.rank, cfg.world_size = device_id, host_threads, rank, world_size
        self.cfg = cfg
        self.ctx = self.lib.fsgpu_create(C.byref(cfg))
        if not self.ctx:
            raise FastoreError("fsgpu_create failed: " + self.lib.fsgpu_create_error().decode())

    def close(self):
        if self.ctx:
            self.lib.fsgpu_destroy(self.ctx)
            self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise FastoreError("fastore_amd error %d: %s" % (rc, self.lib.fsgpu_last_error(self.ctx).decode()))

    @property
    def device_name(self):
        return self.lib.fsgpu_device_name(self.ctx).decode()

    def pack_file(self, in_prefix, out_prefix, verbose=False):
        """`fastore_pack e -i<in_prefix> -o<out_prefix>`"""
        self._check(self.lib.fsgpu_pack_file(self.ctx, in_prefix.encode(), out_prefix.encode(), int(verbose)))
        return self.stats()

    def pack_files(self, in_prefixes, out_prefixes, verbose=False):
        """Several libraries in one go: their bins share the device batches."""
        n = len(in_prefixes)
        a = (C.c_char_p * n)(*[p.encode() for p in in_prefixes]); b = (C.c_char_p * n)(*[p.encode() for p in out_prefixes])
        self._check(self.lib.fsgpu_pack_files(self.ctx, n, a, b, int(verbose)))
        return self.stats()

    def shard_pack(self, in_prefix):
        """code this rank's LPT share of the library's bins and hold the blocks; returns (signatures, own sizes) of the whole
        archive's block table in its final order (numpy uint32 / uint64 arrays; sizes are 0 for the other ranks' blocks)"""
        import numpy as np
        n = C.c_size_t(0)
        self._check(self.lib.fsgpu_shard_pack(self.ctx, in_prefix.encode(), C.byref(n)))
        sigs = np.zeros(n.value, dtype=np.uint32); sizes = np.zeros(n.value, dtype=np.uint64)
        self._check(self.lib.fsgpu_shard_table(self.ctx, sigs.ctypes.data, sizes.ctypes.data, n.value))
        return sigs, sizes

    def shard_write(self, out_prefix, all_sizes):
        """write the held blocks at their offsets (all_sizes = element-wise sum of every rank's size table)"""
        import numpy as np
        a = np.ascontiguousarray(all_sizes, dtype=np.uint64)
        self._check(self.lib.fsgpu_shard_write(self.ctx, out_prefix.encode(), a.ctypes.data, len(a)))

    def shard_pack_set(self, in_prefixes):
        """the same for a set of libraries in ONE device pipeline; returns [(signatures, own sizes)] per library"""
        import numpy as np
        n = len(in_prefixes)
        arr = (C.c_char_p * n)(*[p.encode() for p in in_prefixes]); nb = (C.c_size_t * n)()
        self._check(self.lib.fsgpu_shard_pack_set(self.ctx, n, arr, nb))
        out = []
        for i in range(n):
            sigs = np.zeros(nb[i], dtype=np.uint32); sizes = np.zeros(nb[i], dtype=np.uint64)
            self._check(self.lib.fsgpu_shard_table_of(self.ctx, i, sigs.ctypes.data, sizes.ctypes.data, nb[i]))
            out.append((sigs, sizes))
        return out

    def shard_write_of(self, lib, out_prefix, all_sizes):
        import numpy as np
        a = np.ascontiguousarray(all_sizes, dtype=np.uint64)
        self._check(self.lib.fsgpu_shard_write_of(self.ctx, lib, out_prefix.encode(), a.ctypes.data, len(a)))

    def set_archive_params(self, config, header_fields=b"", quality_codebook=b""):
        """Archive-level parameters for compress_bins(): raw BinModuleConfig, serialized read-id field table, QVZ section."""
        self._check(self.lib.fsgpu_set_archive_params(self.ctx, config, len(config), header_fields or None, len(header_fields)))
        if quality_codebook:
            self._check(self.lib.fsgpu_set_quality_codebook(self.ctx, quality_codebook, len(quality_codebook)))

    def compress_bins(self, batch):
        """The per-bin seam (FastqCompressor::Compress for standard bins): one archive block per bin of `batch`."""
        out = BlockBatch()
        self._check(self.lib.fsgpu_compress_bins(self.ctx, C.byref(batch), C.byref(out)))
        blocks, off = [], 0
        for i in range(out.n_blocks):
            n = out.sizes[i]
            blocks.append(C.string_at(out.data + off, n))
            off += n
        return blocks

    def reset_stats(self):
        self._check(self.lib.fsgpu_reset_stats(self.ctx))

    def stats(self):
        st = Stats()
        self._check(self.lib.fsgpu_get_stats(self.ctx, C.byref(st)))
        return st.as_dict()

    def window_profile(self):
        """phase clocks of the windowed PPMd path (units of 64 shader clocks; zero unless built with -DFS_WIN_PROFILE)"""
        out = (C.c_uint64 * 8)()
        self._check(self.lib.fsgpu_get_window_profile(self.ctx, C.byref(out)))
        d = dict(zip(("fetch", "states_chain", "ranks", "rounds", "writeback", "coder", "windows_total", "streams_total"), list(out)))
        ser = (C.c_uint64 * 2)()
        self._check(self.lib.fsgpu_get_serial_profile(self.ctx, C.byref(ser)))
        d["serial_escapes"], d["serial_update_model"] = ser[0], ser[1]
        return d

    def _encode(self, fn, streams, unit, extra=None):
        n = len(streams)
        ins = (C.c_char_p * n)(*[bytes(s) if len(s) else b"" for s in streams])
        lens = (C.c_size_t * n)(*[len(s) // unit for s in streams])
        caps = [2 * len(s) + 4096 for s in streams]
        bufs = [C.create_string_buffer(c) for c in caps]
        outs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
        capa = (C.c_size_t * n)(*caps)
        outl = (C.c_size_t * n)()
        if extra is None:
            self._check(fn(self.ctx, n, ins, lens, outs, capa, outl))
        else:
            self._check(fn(self.ctx, n, extra, ins, lens, outs, capa, outl))
        return [bufs[i].raw[:outl[i]] for i in range(n)]

    def ppmd_encode(self, streams):
        """PPMd var.J order-4 members, one per input stream (device)."""
        return self._encode(self.lib.fsgpu_ppmd_encode, streams, 1)

    def gather_quality(self, packed, strings):
        """Quality stream of a lossless bin built by fs_gather_quality (device): packed = the stored scores (six bits each,
        MSB first), strings = [(bit offset of the first score, length, emitted back to front?)] in emission order."""
        import numpy as np
        desc = np.zeros(len(strings), dtype=np.dtype([("src_bit", "<u8"), ("len", "<u4"), ("reverse", "<u4")]))
        for i, (b, n, r) in enumerate(strings):
            desc[i] = (b, n, 1 if r else 0)
        total = int(desc["len"].sum())
        out = C.create_string_buffer(total + 16); got = C.c_size_t(0)
        self._check(self.lib.fsgpu_gather_quality(self.ctx, packed, len(packed), desc.ctypes.data, len(strings), out, total + 16, C.byref(got)))
        return out.raw[:got.value]

    def gather_quality_binned(self, packed, bits, binary_threshold, strings):
        """(symbol, context) pairs of an 8-bin (bits=3) / binary (bits=1) quality stream built by fs_gather_quality_pairs (device):
        strings = [(bit offset, length, back to front?, bytes of 'N' positions)] in emission order."""
        import numpy as np
        class S(C.Structure):
            _fields_ = [("src_bit", C.c_uint64), ("len", C.c_uint32), ("reverse", C.c_uint32), ("n_positions", C.c_void_p), ("n_count", C.c_uint32)]
        arr = (S * len(strings))(); keep = []
        total = 0
        for i, (b, n, r, npos) in enumerate(strings):
            buf = C.create_string_buffer(bytes(npos), max(1, len(npos))); keep.append(buf)
            arr[i] = S(b, n, 1 if r else 0, C.cast(buf, C.c_void_p) if len(npos) else None, len(npos)); total += n - len(npos)
        out = C.create_string_buffer(2 * total + 16); got = C.c_size_t(0)
        self._check(self.lib.fsgpu_gather_quality_binned(self.ctx, packed, len(packed), bits, binary_threshold, arr, len(strings), out, total, C.byref(got)))
        return out.raw[:2 * got.value]

    def emit_check(self, in_prefix):
        """(ops expanded, pre-entropy streams compared, streams that differ) between the device's emission kernels and the host's walk over a library."""
        v = [C.c_uint64(0) for _ in range(3)]
        self._check(self.lib.fsgpu_emit_check(self.ctx, in_prefix.encode(), *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def tokeniser_check(self, in_prefix):
        """(read ids tokenised, bins whose IdToken / IdValue streams differ between the device tokeniser and the host's) over a library."""
        r = C.c_uint64(0); d = C.c_uint64(0)
        self._check(self.lib.fsgpu_tokeniser_check(self.ctx, in_prefix.encode(), C.byref(r), C.byref(d)))
        return r.value, d.value

    def matcher_check(self, in_prefix):
        """(reads searched, rows on which the device matcher and the host's window scan disagree) over the standard bins of a library."""
        r = C.c_uint64(0); d = C.c_uint64(0)
        self._check(self.lib.fsgpu_matcher_check(self.ctx, in_prefix.encode(), C.byref(r), C.byref(d)))
        return r.value, d.value

    def unpack_check(self, in_prefix):
        """(plane words compared, words on which fs_unpack_planes -- bases as the bin file packs them -- and fs_pack_bases -- the
        host's unpacked bases -- disagree, reads searched, rows differing from the host scan) over the standard bins of a library."""
        v = [C.c_uint64(0) for _ in range(4)]
        self._check(self.lib.fsgpu_unpack_check(self.ctx, in_prefix.encode(), *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def pe_matcher_check(self, in_prefix):
        """(pairs searched, rows on which the device mate search and the host's disagree) over the standard bins of a paired-end library."""
        r = C.c_uint64(0); d = C.c_uint64(0)
        self._check(self.lib.fsgpu_pe_matcher_check(self.ctx, in_prefix.encode(), C.byref(r), C.byref(d)))
        return r.value, d.value

    def rc_encode(self, models, pair_streams):
        """Range-coded streams; pair_streams[i] = interleaved (symbol, ctx0) bytes, models[i] in 0..5."""
        m = (C.c_uint32 * len(models))(*models)
        return self._encode(self.lib.fsgpu_rc_encode, pair_streams, 2, extra=m)

    def qvz_encode(self, footer, blocks):
        """QVZ (--lossy) quality streams; footer = the .bmeta quality section (WELL state, max read length, codebook),
        blocks[i] = (read_lens uint32 array, quality values uint8 array) of one block in coding order."""
        import numpy as np
        n = len(blocks)
        lens = [np.ascontiguousarray(b[0], dtype=np.uint32) for b in blocks]
        quals = [np.ascontiguousarray(b[1], dtype=np.uint8) for b in blocks]
        for l, q in zip(lens, quals):
            if int(l.sum()) != len(q):
                raise FastoreError("read lengths do not add up to the number of quality values")
        qp = (C.c_void_p * n)(*[q.ctypes.data for q in quals]); lp = (C.c_void_p * n)(*[l.ctypes.data for l in lens])
        nr = (C.c_size_t * n)(*[len(l) for l in lens])
        caps = [3 * len(q) + 64 for q in quals]
        bufs = [C.create_string_buffer(c) for c in caps]
        outs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
        capa = (C.c_size_t * n)(*caps); outl = (C.c_size_t * n)()
        self._check(self.lib.fsgpu_qvz_encode(self.ctx, bytes(footer), len(footer), n, qp, lp, nr, outs, capa, outl))
        return [bufs[i].raw[:outl[i]] for i in range(n)]
