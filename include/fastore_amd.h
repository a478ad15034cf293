/* fastore_amd -- C ABI of the MI355X-native fastore_pack hot path.
 *
 * The reference (refresh-bio/FaStore v0.8.0) has no plugin API; the seam this library replaces is
 * the per-bin call
 *     FastqCompressor::Compress(reads, packCtx, signature, rawDnaSize, workBin, compBin)
 *         fastore/fastore_pack/FastqCompressor.h:1215-1220, FastqCompressor.cpp:5689-5718
 * and, one level up, the body of `fastore_pack e`
 *     CompressorModuleSE/PE::Bin2Dnarch
 *         fastore/fastore_pack/CompressorModule.h:29-49, CompressorModule.cpp:34-454, 599-1091.
 * Plain pointers and sizes only; no exceptions cross this boundary; every function returns 0 on
 * success or a negative code, and fsgpu_last_error() gives the text that the reference would have
 * printed after "Error: " (fastore_pack/main.cpp:117-121).
 * One context per GPU; a context is not re-entrant (like one reference compressor instance per
 * worker thread, fastore_pack/CompressorOperator.h:45-59).
 */
#ifndef FASTORE_AMD_H
#define FASTORE_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSGPU_OK 0
#define FSGPU_ERR_ARG (-1)
#define FSGPU_ERR_DEVICE (-2)   /* no usable HIP device / kernel failure: the product has no CPU fallback */
#define FSGPU_ERR_FORMAT (-3)
#define FSGPU_ERR_IO (-4)
#define FSGPU_ERR_INTERNAL (-5)

typedef struct fsgpu_ctx fsgpu_ctx;

/* fastore_pack command-line knobs (fastore_pack/main.cpp:165-301, Params.h:18-147).
 * Zero-initialise, then call fsgpu_config_defaults(). */
typedef struct fsgpu_config {
    uint32_t min_bin_size;              /* -f  (default 256: BinExtractorParams) */
    int32_t encode_threshold;           /* -e  0 = auto */
    int32_t pair_encode_threshold;      /* -E  0 = auto */
    int32_t shift_cost;                 /* -s */
    int32_t mismatch_cost;              /* -m */
    uint32_t max_lz_window;             /* -w */
    uint32_t max_pair_lz_window;        /* -W */
    uint32_t extra_reduce_hard_reads;   /* -r */
    uint32_t extra_reduce_expensive_lz; /* -l */
    uint32_t max_record_shift_diff;     /* -q */
    uint32_t max_new_variants_per_read; /* -n */
    uint32_t max_hamming_distance;      /* -d */
    uint32_t min_consensus_size;        /* -c */
    int32_t device_id;                  /* HIP device ordinal */
    uint32_t host_threads;              /* worker threads of the host stages (0 = all cores) */
    uint32_t max_waves;                 /* resident coder wavefronts (0 = 12 per CU = 3 per SIMD, memory permitting) */
    uint64_t batch_bases;               /* bases per device batch (0 = default) */
    uint32_t rank, world_size;          /* bin sharding: this context packs its share of the standard bins -- longest-processing-time-first over the .bmeta record totals, the same table on every rank (packer.cpp: shardOwners); block 0 is rank 0's */
    uint32_t pipeline_slices;           /* slices a batch is cut into so that host front end and device overlap (0 = default 8, 1 = off) */
    uint32_t pipeline_lanes;            /* engine instances (HIP streams) whose kernels may overlap on the GPU (0 = one per slice, at most 8) */
    uint32_t one_shot;       /* 1: the context packs once and is destroyed (the CLI): its device is made on a thread of its own (a missing
                                device is reported by the first call that needs it), buffers are not pre-sized for a next batch */
    uint32_t reserved1;
} fsgpu_config;

/* Unpacked reads of a batch of bins, structure-of-arrays (what the reference hands to Compress() as
 * vector<FastqRecord> + PackContext; record/graph grammar: fastore_rebin/NodesPacker.cpp:567-979).
 * bases/quals share offsets; a paired record stores mate 1 then mate 2 contiguously. */
typedef struct fsgpu_record {
    uint32_t seq_off, head_off;
    uint16_t seq_len, aux_len, minim_pos;
    uint8_t head_len, flags;            /* bit0 reverse-complemented, bit1 pair swapped */
} fsgpu_record;
typedef struct fsgpu_node { uint32_t rec, em_begin, em_count, tree_begin, tree_count; } fsgpu_node;
typedef struct fsgpu_tree { uint32_t signature; int32_t main_signature_pos; uint32_t node_begin, node_count; } fsgpu_tree;
typedef struct fsgpu_bin {
    uint32_t signature, min_len, max_len;
    uint64_t raw_dna_size;
    uint32_t rec_begin, rec_count, top_begin, top_count;
} fsgpu_bin;
typedef struct fsgpu_bin_batch {
    const uint8_t *bases, *quals, *heads;
    size_t n_bases, n_heads;
    const fsgpu_record* records; size_t n_records;
    const fsgpu_node* nodes; size_t n_nodes;
    const uint32_t* top_nodes; size_t n_top_nodes;
    const uint32_t* em_records; size_t n_em_records;
    const fsgpu_tree* trees; size_t n_trees;
    const fsgpu_bin* bins; size_t n_bins;
} fsgpu_bin_batch;

/* Compressed blocks of a batch; memory is owned by the context and valid until the next call. */
typedef struct fsgpu_block_batch {
    const uint8_t* data;                /* blocks back to back, in bin order */
    const uint64_t* sizes;              /* n_blocks entries */
    size_t n_blocks;
} fsgpu_block_batch;

typedef struct fsgpu_stats {
    double encode_kernel_ms, assemble_kernel_ms;   /* HIP-event time on the engine's stream */
    double frontend_ms, block0_ms, io_ms, total_ms;
    uint64_t kernel_launches, stream_items, ppmd_symbols, rc_symbols, ppmd_restarts;
    uint64_t h2d_bytes, d2h_bytes;
    uint64_t bins, records, algorithmic_bytes;      /* 2*(seq+aux)+head per record + block bytes */
    uint64_t block0_records, block0_bytes, cdata_bytes;
    /* always 0: no standard-bin stream is ever coded on the host (there is no CPU fallback for the device stages; the fields stay
     * for the layout's sake and so that a caller can assert it) */
    uint64_t host_coded_symbols, host_coded_streams;
    /* windowed PPMd hit path (ppmd_window.h): window attempts, windows coded, symbols coded inside windows (of
     * ppmd_symbols), rounds (positions sharing a context are processed one rank per round), windows redone shorter */
    uint64_t ppmd_window_attempts, ppmd_windows, ppmd_window_symbols, ppmd_window_rounds, ppmd_windows_redone;
    uint64_t ppmd_window_light_rounds;   /* of ppmd_window_rounds: rounds that needed no swap and no rescale */
    /* fs_gather_quality (device-side quality path of lossless archives): HIP-event time, scores gathered, bytes read + written */
    double gather_kernel_ms; uint64_t gather_symbols, gather_bytes;
    /* device-side read matcher (matcher.hip): reads searched, time the host threads spent in its calls (summed over the
     * threads: upload, kernels, download, waiting), HIP-event time of its kernels (summed over the calls) */
    uint64_t matcher_reads; double matcher_call_ms, matcher_kernel_ms;
    uint64_t tokenised_ids;              /* read ids split into the IdToken / IdValue streams by fs_tokenise_ids (device) */
    uint64_t device_batches;             /* device batches the standard bins were cut into (fsgpu_config.batch_bases each at most) */
    uint64_t ppmd_max_restarts;          /* most model restarts (sub-allocator exhausted, Model.cpp:109-140 again) inside ONE PPMd stream */
    uint64_t stolen_bins;                /* always 0 since round 5 (the work-stealing tail of bin-sharded packs left the tree; the field stays for the layout's sake) */
    /* window search (fs_match_reads): bytes of bases that went up for it -- packed as the bin file stores them (.bdna) plus a
     * descriptor per read when the device unpacks them itself (fs_unpack_planes; FastqPacker.cpp:290-411), else ASCII --
     * and the reads whose bases the device unpacked */
    uint64_t matcher_bases_h2d_bytes, matcher_unpacked_reads;
    /* mate search of paired-end bins on the device (fs_match_mates; FS_DEVICE_MATES=1: a bin at a time, 2: many bins a launch, nobody waits):
     * pairs searched, wall time of its calls (summed over the callers), HIP-event time of its kernels (summed over the launches) */
    uint64_t mate_pairs; double mate_call_ms, mate_kernel_ms;
    /* From here on the layout is APPEND-ONLY: new counters go behind the last one, none is taken out (struct_bytes says how much
     * of the struct the library filled, so a caller built against an older header reads what it knows and nothing shifted). */
    uint64_t struct_bytes;               /* sizeof(fsgpu_stats) of the library that filled it */
    uint64_t ppmd_window_drops;          /* rescales inside windows that let states drop out of their context (units shrunk at the window's commit; rounds 2-4 ended the window there) */
    uint64_t coder_tail_launches;        /* coder launches made again for the streams a launch left in its queue: its workgroups found no free arena slot (none waits inside a kernel) or had coded their share of one launch */
} fsgpu_stats;

void fsgpu_config_defaults(fsgpu_config* cfg);
int fsgpu_device_count(void);
fsgpu_ctx* fsgpu_create(const fsgpu_config* cfg);          /* NULL when no HIP device is usable; see fsgpu_create_error() */
const char* fsgpu_create_error(void);
void fsgpu_destroy(fsgpu_ctx* ctx);
const char* fsgpu_last_error(const fsgpu_ctx* ctx);

/* Archive-level parameters that travel inside .bmeta: the raw 88-byte BinModuleConfig
 * (fastore_bin/Params.h:167-193) and the serialized read-id field table exactly as stored in the
 * .bmeta footer (fastore_bin/BinFile.cpp:395-461; may be NULL/0 for header-less archives). */
int fsgpu_set_archive_params(fsgpu_ctx* ctx, const void* bin_module_config, size_t config_bytes,
                             const uint8_t* header_fields, size_t header_fields_bytes);

/* --lossy (QVZ) libraries only: the quality section of the .bmeta footer -- WELL-1024a state (128 bytes),
 * max_read_length (u32), codebook -- exactly as BinFileWriter stores it (fastore_bin/BinFile.cpp:386-394,
 * fastore_bin/QVZ.cpp:165-222).  Call after fsgpu_set_archive_params(); trailing bytes are ignored. */
int fsgpu_set_quality_codebook(fsgpu_ctx* ctx, const uint8_t* qvz_footer, size_t qvz_footer_bytes);

/* The per-bin hot path on standard (LZ) bins: FastqCompressor::Compress for signature != 4^p. */
int fsgpu_compress_bins(fsgpu_ctx* ctx, const fsgpu_bin_batch* in, fsgpu_block_batch* out);

/* Entropy-coder entry points (one call = many independent streams on the device).
 * PPMd var.J member as PpmdEncoder::EncodeNextMember emits it (ppmd/PPMd.cpp:44-70,119-154);
 * range-coded stream of a TEncoder<model> (rc/ContextEncoder.h:208-250) over (symbol, ctx0) pairs;
 * model: 0 <2,4> simple, 1 <8,4> simple, 2 <8,4> adv, 3 <2,10> adv, 4 <8,6> adv, 5 <256,1> adv. */
int fsgpu_ppmd_encode(fsgpu_ctx* ctx, size_t n_streams, const uint8_t* const* in, const size_t* in_len,
                      uint8_t* const* out, const size_t* out_cap, size_t* out_len);
int fsgpu_rc_encode(fsgpu_ctx* ctx, size_t n_streams, const uint32_t* model, const uint8_t* const* pairs,
                    const size_t* n_pairs, uint8_t* const* out, const size_t* out_cap, size_t* out_len);

/* QVZ quality stream of one block per stream: QVZEncoder (fastore_pack/qv_compressor.h:127-160, arith.cpp:33-125,
 * qv_stream.cpp:19-71) driven by IQualityStoreBase::CompressReadQuality's MET_QVZ loop (FastqCompressor.cpp:318-364)
 * with the WELL generator reset at the block start.  quals[i]: the block's quality values (offset removed, < 72),
 * read after read in coding order; read_lens[i][n_reads[i]].  The model is parsed from `qvz_footer` as above. */
int fsgpu_qvz_encode(fsgpu_ctx* ctx, const uint8_t* qvz_footer, size_t qvz_footer_bytes, size_t n_streams,
                     const uint8_t* const* quals, const uint32_t* const* read_lens, const size_t* n_reads,
                     uint8_t* const* out, const size_t* out_cap, size_t* out_len);

/* Quality stream of a lossless bin, built on the device (fs_gather_quality): what FastqPacker's six-bit unpack
 * (fastore_bin/FastqPacker.cpp:290-411, scores stored MSB first) followed by IQualityStoreBase::CompressReadQuality for
 * MET_NONE (fastore_pack/FastqCompressor.cpp:229-247: score minus the archive's offset, back to front for a record stored
 * reverse-complemented) leave in the PPMd input buffer.  strings[i]: bit offset of the i-th emitted read's first stored
 * score in `packed`, its length, and whether it is emitted back to front; out receives the strings one after the other.
 * Timing of the last call: fsgpu_stats.gather_kernel_ms / gather_symbols / gather_bytes. */
typedef struct fsgpu_quality_string { uint64_t src_bit; uint32_t len; uint32_t reverse; } fsgpu_quality_string;
int fsgpu_gather_quality(fsgpu_ctx* ctx, const uint8_t* packed, size_t packed_bytes, const fsgpu_quality_string* strings,
                         size_t n_strings, uint8_t* out, size_t out_cap, size_t* out_len);

/* The same for 8-bin (`bits` = 3) and binary (`bits` = 1) archives (MET_8BIN / MET_BINARY, FastqCompressor.cpp:249-316): the
 * stream is (symbol, context) byte pairs for the order-k range coder, positions under an 'N' base (n_positions: stored
 * indices inside the string) are left out; binary_threshold maps a stored bit (score 6 / 40) to the coded symbol.
 * out receives the pairs (2 bytes each), *out_pairs their number. */
typedef struct fsgpu_quality_string_n { uint64_t src_bit; uint32_t len; uint32_t reverse; const uint8_t* n_positions; uint32_t n_count; } fsgpu_quality_string_n;
int fsgpu_gather_quality_binned(fsgpu_ctx* ctx, const uint8_t* packed, size_t packed_bytes, uint32_t bits, uint32_t binary_threshold,
                                const fsgpu_quality_string_n* strings, size_t n_strings, uint8_t* out, size_t out_cap_pairs, size_t* out_pairs);

/* Parity check of the device-side read matcher (matcher.hip; ReadsClassifierSE::ConstructMatchTree's window search,
 * fastore_pack/ReadsClassifier.cpp:55-83, 95-442): every standard bin of the library <in_prefix> goes through the host's
 * serial window scan and through the device; *reads = reads searched, *differing = rows (matched read, cost, shift,
 * mismatch-free flag, exact-duplicate flag) on which the two disagree. */
int fsgpu_matcher_check(fsgpu_ctx* ctx, const char* in_prefix, uint64_t* reads, uint64_t* differing);

/* Parity check of the device-side unpack of the bases (matcher.hip: fs_unpack_planes; the reader side of the bin file,
 * IFastqPacker::ReadNextRecord, fastore_bin/FastqPacker.cpp:290-411, and the signature the packer leaves out): every standard bin
 * of the library <in_prefix> goes through the window search with its bases as the bin file stores them (.bdna bytes + a descriptor
 * per read) while the same bit planes are built from the bases the host unpacked; *plane_words = words compared, *differing =
 * words on which the two disagree; *reads / *differing_rows as fsgpu_matcher_check (the rows of the search on the device-unpacked
 * bases against the host scan). */
int fsgpu_unpack_check(fsgpu_ctx* ctx, const char* in_prefix, uint64_t* plane_words, uint64_t* differing, uint64_t* reads, uint64_t* differing_rows);

/* Parity check of the device-side mate search of paired-end bins (matcher.hip: fs_match_mates; LzCompressorPE::CompressPair's
 * history search, fastore_pack/FastqCompressor.cpp:4610-4959, with the minimizer sets of FastqCategorizerBase::FindMinimizers,
 * fastore_bin/FastqCategorizer.cpp:109-151): every standard bin of the paired-end library <in_prefix> goes through the host's
 * serial search and through the device; *pairs = pairs searched, *differing = rows (cost, shift, mismatch-free flag, matched
 * pair, its place in the history) on which the two disagree.  A single-end library gives 0 / 0. */
int fsgpu_pe_matcher_check(fsgpu_ctx* ctx, const char* in_prefix, uint64_t* pairs, uint64_t* differing);

/* Parity check of the device-side read-id tokeniser (fs_tokenise_ids; IHeaderStoreBase::CompressReadId, fastore_pack/
 * FastqCompressor.cpp:504-583): every standard bin's read ids, in stored order, through the host's restatement (unpacked
 * headers) and through the kernel (packed .bhead bytes); *ids = read ids tokenised, *differing_bins = bins whose IdToken or
 * IdValue pair stream differs. */
int fsgpu_tokeniser_check(fsgpu_ctx* ctx, const char* in_prefix, uint64_t* ids, uint64_t* differing_bins);

/* Parity check of the device-side stream emission (fs_emit_count / fs_emit_scan / fs_emit_write, fs_rle_binary, fs_rle0): the
 * streams of a bin that hold bases -- HardReads, LettersX, Match, MatchBinary, CMatch, CLetters, their paired-end counterparts --
 * and the run-length coded LZ ids (CompressHardRead / CompressNormalMatch / CompressContigRead / StoreContigDefinition,
 * fastore_pack/FastqCompressor.cpp:1388-1760; LzCompressorPE::CompressPair :4740-4900; BinaryRleEncoder / Rle0Encoder,
 * rle/RleEncoder.h:21-79, 140-212).  Every standard bin through the host's walk twice -- writing those streams itself, and leaving
 * ops for the device -- and the ops through the kernels: *ops = ops expanded, *streams = streams compared byte for byte (pre-entropy),
 * *differing = streams that differ (every other stream of the bin must be the walk's own either way and counts here too). */
int fsgpu_emit_check(fsgpu_ctx* ctx, const char* in_prefix, uint64_t* ops, uint64_t* streams, uint64_t* differing);

/* Whole `fastore_pack e -i<in_prefix> -o<out_prefix>`: reads .bmeta/.bdna/.bqua/.bhead, writes
 * .cmeta/.cdata in the reference's -t1 block order. */
int fsgpu_pack_file(fsgpu_ctx* ctx, const char* in_prefix, const char* out_prefix, int verbose /* 0 quiet, 1 = -v, 2 = progress line only */);

/* The same for n libraries at once: their bins share the device batches (more independent streams in
 * flight per launch), each library gets its own archive. */
int fsgpu_pack_files(fsgpu_ctx* ctx, size_t n, const char* const* in_prefixes, const char* const* out_prefixes, int verbose);

/* Adapter for callers (and tests) that do not hold unpacked records yet: every standard bin (signature != 4^p,
 * >= min_bin_size records) of a binned library as one flat batch, in ascending signature order -- what the reference's
 * BinFileExtractor::ExtractNextStdBin + IFastqNodesPacker::UnpackFromBin produce bin by bin
 * (fastore_pack/BinFileExtractor.cpp:21-103, fastore_rebin/NodesPacker.cpp:416-679) -- together with the archive-level
 * parameters of the .bmeta footer in the form fsgpu_set_archive_params()/fsgpu_set_quality_codebook() take.
 * No device needed.  fsgpu_library_open returns NULL on error (message: fsgpu_create_error()). */
typedef struct fsgpu_library fsgpu_library;
fsgpu_library* fsgpu_library_open(const char* in_prefix, uint32_t min_bin_size);
void fsgpu_library_close(fsgpu_library* lib);
const fsgpu_bin_batch* fsgpu_library_std_bins(const fsgpu_library* lib);
const void* fsgpu_library_config(const fsgpu_library* lib, size_t* bytes);                 /* raw BinModuleConfig */
const uint8_t* fsgpu_library_header_fields(const fsgpu_library* lib, size_t* bytes);       /* NULL/0 without read ids */
const uint8_t* fsgpu_library_quality_codebook(const fsgpu_library* lib, size_t* bytes);    /* NULL/0 unless --lossy */

/* Bin-sharded packing of ONE library by `world_size` contexts (one per GPU; fsgpu_config.rank / world_size) WITHOUT part
 * files.  Every context takes its share of the standard bins -- longest-processing-time-first over the per-signature
 * record totals of the .bmeta footer (fastore_bin/BinFile.h:57-79), a pure function of the footer, so nothing is exchanged
 * for it; the merged small-bins/N block goes to rank 0 -- in three steps:
 *   fsgpu_shard_pack   codes the rank's bins and holds the blocks in memory; *n_blocks = blocks of the WHOLE archive
 *   fsgpu_shard_table  the archive's block table in its final (-t1) order: signatures (the same on every rank) and this
 *                      rank's block sizes, 0 for the blocks of other ranks
 *   (caller)           element-wise sum of the size tables of all ranks: ONE all-reduce (or all-gather) of n_blocks u64
 *                      over RCCL between processes (fastore_amd/shard.py), a plain sum inside one process (-G<n>)
 *   fsgpu_shard_write  writes the held blocks at their offsets in <out_prefix>.cdata; rank 0 also writes the .cmeta
 * No block bytes cross ranks; the archive equals the single-GPU (and the reference's -t1) archive byte for byte. */
int fsgpu_shard_pack(fsgpu_ctx* ctx, const char* in_prefix, size_t* n_blocks);
int fsgpu_shard_table(const fsgpu_ctx* ctx, uint32_t* signatures, uint64_t* sizes, size_t n_blocks);
int fsgpu_shard_write(fsgpu_ctx* ctx, const char* out_prefix, const uint64_t* all_sizes, size_t n_blocks);
/* The same for a SET of libraries whose bins share one device pipeline (a rank's share of every library is coded side by
 * side: the unit of sharding stays the bin): n_blocks[i] / lib index the libraries in the order of in_prefixes. */
int fsgpu_shard_pack_set(fsgpu_ctx* ctx, size_t n, const char* const* in_prefixes, size_t* n_blocks);
int fsgpu_shard_table_of(const fsgpu_ctx* ctx, size_t lib, uint32_t* signatures, uint64_t* sizes, size_t n_blocks);
int fsgpu_shard_write_of(fsgpu_ctx* ctx, size_t lib, const char* out_prefix, const uint64_t* all_sizes, size_t n_blocks);

/* The file-based form of the same: bin-sharded packing of one library by `world_size` contexts (one per GPU; fsgpu_config.rank / world_size): every
 * context writes <out_prefix>.part<rank>.{cdata,cmeta}; this call merges the parts into <out_prefix>.{cdata,cmeta} in
 * the reference's -t1 order (the merged small-bins/N block of rank 0 first, then ascending signature) and removes them.
 * Host only.  The multi-process form of the same exchange (all-gather of sizes over RCCL) is fastore_amd/shard.py.
 * Returns 0 or a negative FSGPU_ERR_*; the message goes to `err` (may be NULL). */
int fsgpu_merge_parts(const char* out_prefix, uint32_t world_size, char* err, size_t err_len);

/* The reference's -v statistics (`StreamSizes:` ... on stdout, fastore_pack/CompressorModule.cpp:357-387) of a finished
 * archive <out_prefix>.{cmeta,cdata}, summed from the block headers -- what `fastore_pack e -v -G<n>` prints after the
 * merge.  Host only. */
int fsgpu_print_stream_sizes(const char* out_prefix, char* err, size_t err_len);

int fsgpu_get_stats(const fsgpu_ctx* ctx, fsgpu_stats* out);   /* cumulative since fsgpu_reset_stats() */
int fsgpu_reset_stats(fsgpu_ctx* ctx);
/* Diagnostic: the per-phase clock sums of the windowed PPMd path since fsgpu_reset_stats(), in units of 64 shader clocks
 * (all zero unless the library was built with -DFS_WIN_PROFILE): [0] input/hint/record fetch, [1] state lists + chain,
 * [2] ranks, [3] rounds, [4] write-back, [5] range coder, [6] whole windows, [7] whole PPMd streams. */
int fsgpu_get_window_profile(const fsgpu_ctx* ctx, uint64_t out[8]);
/* -DFS_WIN_PROFILE builds: clocks / 64 of the serial path spent in escapes (suffix walk + masked contexts) and in UpdateModel */
int fsgpu_get_serial_profile(const fsgpu_ctx* ctx, uint64_t out[2]);
const char* fsgpu_device_name(const fsgpu_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
