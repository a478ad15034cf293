// Plain descriptors shared by the host engine and the HIP kernels.
#pragma once
#include <stdint.h>

namespace fsdev {

enum : uint32_t { KIND_PPMD = 0, KIND_RC_BASE = 1 /* + fsrc::Model */, KIND_QVZ = 64 /* fsqvz arithmetic coder */ };

// one entropy-coded stream of one bin
struct StreamItem {
    uint64_t in_off;      // byte offset into the batch input buffer (even for RC pair streams)
    uint64_t out_off;     // byte offset into the batch scratch-output buffer
    uint32_t in_len;      // bytes (PPMd), (symbol, ctx) pairs (RC) or u32 symbols (QVZ)
    uint32_t out_cap;     // bytes available at out_off
    uint32_t kind;        // KIND_PPMD, KIND_RC_BASE + model, or KIND_QVZ
    uint32_t bin;         // bin index inside the batch
    uint64_t aux_off;     // KIND_QVZ: byte offset of the library's model blob in the batch input buffer (16-byte aligned)
};

// fs_gather_quality: one quality string of a slice, in emission order.  src_bit: bit offset of its first stored score in
// the batch input buffer (six bits per score, MSB first); dst_off: byte offset of its first coded symbol in the gather
// region behind the uploaded input; reverse: the string is emitted back to front
struct QuaString { uint64_t src_bit; uint32_t dst_off; uint16_t len; uint16_t reverse; };
// device-only part of a batch input buffer: quality streams gathered on the device instead of uploaded
// 8-bin / binary archives: the stream is (symbol, context) byte pairs, symbol = the stored 3- or 1-bit score, context =
// (emitted index * 8 or 2) / length; the n_count positions listed (stored indices, n_off into the slice's list) lie under
// an 'N' base and are left out.  dst_off counts PAIRS.
struct QuaPairString { uint64_t src_bit; uint32_t dst_off, n_off; uint16_t len; uint8_t reverse, n_count; };
// --lossy (QVZ) archives, fs_gather_quality_qvz: the stream is one u32 per score, context | state << 24 -- the context is the
// conditional quantizer chosen for the score (column, previous quantized value, the low or the high one by a 7-bit draw of the
// archive's WELL-1024a generator), the state the place of the quantized value in that quantizer's output alphabet
// (IQualityStoreBase::CompressReadQuality, MET_QVZ: fastore_pack/FastqCompressor.cpp:318-364; choose_quantizer, quantizer.cpp:522-531).
// dst_off counts BYTES in the gather region; draw0 = scores of the bin's stream in front of this string (the generator is
// re-seeded per bin and draws once per score, in emission order); model16 = place of the library's QvzSymHeader in the batch
// input, in 16-byte units
struct QuaQvzString { uint64_t src_bit; uint32_t dst_off, draw0, model16; uint16_t len, reverse; };
// The tables of one library's conditional quantizers as fs_gather_quality_qvz reads them (built by fs::QvzModel::parse from the
// codebook in .bmeta; byte offsets from the header's start): col_ctx_base u32[columns]; col_index u16[columns][82] (previous
// quantized value -> index in the column's input alphabet, 0xFFFF absent); qratio u8[n_ctx / 2]; quant, state_of u8[n_ctx][72]
// (score -> quantized value / its place in the output alphabet, 0xFF unused); well u32[well_words]: the generator's outputs from
// the archive's seed on -- four 7-bit draws per word, the word's top four bits unused (well_1024a_bits, well.cpp:42-55)
struct QvzSymHeader { uint32_t columns, n_ctx, col_ctx_base_off, col_index_off, qratio_off, quant_off, state_of_off, well_off, well_words, total_bytes, pad[2]; };
// bits: 6 = QuaString descriptors and byte output (qvz: QuaQvzString descriptors and u32 output); 3 / 1 = QuaPairString descriptors, pair output, n_list_off = the 'N' positions
struct GatherPlan { uint64_t desc_off = 0; uint32_t n_strings = 0; uint64_t out_bytes = 0; uint64_t symbols = 0; uint32_t bits = 6; uint64_t n_list_off = 0, n_list_bytes = 0;
                    uint32_t sym_of_bit[2] = {0, 1};      // binary archives: the coded symbol of a stored 0 / 1 (score 6 / 40 against the archive's threshold)
                    uint32_t qvz = 0; };                  // bits == 6 and qvz: QuaQvzString descriptors, fs_gather_quality_qvz

// Device-side read matcher (matcher.hip): the reads of a bin's match-tree constructions, each construction's reads in
// processing order; a row of answers per read
struct MatchRead { uint32_t seq_off; uint16_t len, min_pos; };            // bases (ASCII) at seq[seq_off .. +len); signature position
// The same read as the bin file stores it (.bdna, fastore_bin/FastqPacker.cpp:290-411): its bases from bit `bit_off` of the bin's
// packed bytes on, MSB first, two bits each (index into the archive's symbol order) or three when the read holds an 'N' -- all
// but the bases of the signature, which are not stored: they stand at sig_pos and are the digits of sig_id (base 4, first base
// in the highest digit).  info = sig_id (bits 0-21) | sig_pos << 22 (8 bits) | two-bit form << 30 | has a signature << 31.
struct PackedRead { uint32_t bit_off, info; };
enum : uint32_t { PACKED_SIG_BITS = 22u, PACKED_PLAIN = 1u << 30, PACKED_HAS_SIG = 1u << 31 };
// a bin's packed bases for the matcher: reads[i] describes MatchRead i
struct PackedDna { const uint8_t* dna; size_t bytes; const PackedRead* reads; uint8_t symbol_order[8]; uint32_t sig_len; };
// reads[first .. first+count) in processing order; aux: read holding the sub-tree's root copy, or -1; warm: the window as it
// stands in front of reads[first] when this is a later piece of a long construction -- warm_count reads, oldest first, named
// in the warm-up list from warm_first on (a piece's answers do not depend on how the construction was cut)
struct MatchCall { uint32_t first, count; int32_t aux; uint32_t warm_first, warm_count, pad; };
struct MatchParams { uint32_t window; int32_t shift_cost, mismatch_cost, encode_threshold; };   // -w, -s, -m, -e (0 = read length / 2)
// match: read the best window slot holds (-1 none at or below the threshold, -2 a dummy slot); cost / shift / no_mismatches of
// that match (cost = threshold + 1 when none); identical: exact duplicate of a slot that is not the root copy
struct MatchRow { int32_t match; int16_t cost, shift; uint8_t no_mismatches, identical, dummy, pad; };

// Device-side mate search of paired-end bins (matcher.hip: fs_match_mates; LzCompressorPE::CompressPair, fastore_pack/
// FastqCompressor.cpp:4610-4959).  A bin's pairs in the order the tree walk emits them: where the mate's bases are, how many,
// and its encode threshold (seqLen / 1.5 in double, truncated -- computed by the host: the device does no floating point).
struct MatePair { uint32_t mate_off; uint16_t mate_len; int16_t threshold; };
struct MateJob { uint32_t first, count; };
// -W (history entries), -s, -m; the archive's minimizer parameters (signature length <= 8: one bit per signature in LDS)
struct MateParams { uint32_t window; int32_t shift_cost, mismatch_cost; uint32_t sig_len, skip_zone; uint8_t symbol_order[8]; };
// cost 255: no candidate at all.  match: index (in the job's pair list) of the pair whose mate sits in the matched history
// entry, valid when cost <= threshold; prev_id: that entry's place in the history at the time of the search
struct MateRow { int32_t match; int16_t cost, shift; uint16_t prev_id; uint8_t no_mismatches, overflow; };

// Device-side read-id tokeniser (fs_tokenise_ids).  Field table of a library, as the kernel reads it: n_fields, then per
// field an IdField, then the token fields' value lists (per value: u32 offset from the blob start, u32 length), then bytes
struct IdField { uint8_t separator, is_const, is_numeric, plog; uint32_t n_values, values_off; uint64_t min_value; };
struct IdString { uint64_t src_bit; uint32_t len, pad; };      // first stored character (7 bits each, after the implied '@'); length incl. the '@'
// one bin: its strings, where its two streams go (byte offsets in the gather region), the items whose in_len the kernel sets
struct IdJob { uint32_t first, count, tok_item, val_item; uint64_t tok_out, val_out, table_off; };
struct IdPlan { uint64_t jobs_off = 0, strings_off = 0; uint32_t n_jobs = 0, n_strings = 0; uint64_t out_bytes = 0; };

// Device-side stream emission (fs_emit_count / fs_emit_scan / fs_emit_write, fs_rle_binary, fs_rle0: SURVEY 8 a6 + a11).  The host's walk
// over a bin's match trees decides WHAT is coded (CompressNode ... CompressSubTree, fastore_pack/FastqCompressor.cpp:1279-2119) and
// leaves one op per record whose coding reads bases; the device compares the bases and writes the streams that hold them:
//   HARD      a read without a match: its bases around the signature              -> HardReads            (CompressHardRead, :1388-1410)
//   MATCH     a read against its LZ match at a shift: the overhangs' letters, per overlapping position (bar the signature) a match
//             bit (run-length coded: BinaryRleEncoder) or a match symbol, a letter with the matched base as context per mismatch
//                                                                                 -> LettersX, Match | MatchBinary (CompressNormalMatch, :1460-1560)
//   CREAD     a read of a contig against the consensus: the cut zones in full, elsewhere the variant positions
//                                                                                 -> CLetters             (CompressContigRead, :1690-1760)
//   CDEF      a contig's definition: a bit per position (variant or not, run-length coded), the consensus letters of the ends
//             and the variants                                                    -> CMatch, CLetters     (StoreContigDefinition, :1620-1680)
//   PE_HARD / PE_MATCH  the mate of a pair, likewise                              -> HardPE | LettersXPE, MatchRlePE | MatchBinaryPE
//                                                                                    (LzCompressorPE::CompressPair, :4740-4900)
// and the LZ ids of a bin (one per MATCH, in order) go through Rle0Encoder on the device                   -> LzId
// Offsets seq_a / seq_b count from the bin's first base in the slice's base region; a contig's bytes (sequence[2 L], variant[2 L])
// from the bin's first contig byte.
enum : uint32_t { EMIT_HARD = 1, EMIT_MATCH = 2, EMIT_CREAD = 3, EMIT_CDEF = 4, EMIT_PE_HARD = 5, EMIT_PE_MATCH = 6 };
enum : uint32_t { EMIT_SHIFT_ONLY = 0, EMIT_FULL = 1, EMIT_EXPENSIVE = 2 };
// channels: what the ops write.  L = bytes or (symbol, context) pairs, B = bits (one byte each, for the run-length coder) or match symbols
enum : uint32_t { ECH_HARD = 0, ECH_LETTERS = 1, ECH_MATCH_BITS = 2, ECH_MATCH_BIN = 3, ECH_CMATCH_BITS = 4, ECH_CLETTERS = 5,
                  ECH_HARD_PE = 6, ECH_LETTERS_PE = 7, ECH_MATCH_BITS_PE = 8, ECH_MATCH_BIN_PE = 9, ECH_COUNT = 10 };
struct EmitOp {
    uint32_t kind;            // EMIT_*
    uint32_t seq_a;           // the read's (mate's) bases
    uint32_t seq_b;           // MATCH / PE_MATCH: the matched read's (mate's) bases; CREAD / CDEF: the contig's bytes
    uint16_t len_a, len_b;    // lengths of a and b; CDEF: rangeFirst, rangeSecond
    uint16_t pos_a, pos_b;    // HARD / CREAD: the read's signature position; MATCH: the matched read's; CDEF: the main read's; (CREAD) pos_b = the contig's read length
    int16_t shift;            // MATCH / PE_MATCH
    uint8_t mode;             // MATCH / PE_MATCH: EMIT_SHIFT_ONLY / EMIT_FULL / EMIT_EXPENSIVE
    uint8_t pad;
    uint32_t pad2[2];
};
// one bin's emission: its ops, its LZ ids, where its bases and contig bytes stand (byte offsets in the batch input), the stream items
// of the eleven streams the device writes (their in_len is set by the kernels) and where each stream goes (byte offsets in the
// device-only region behind the input).  raw_off[c]: scratch space of the run-length coded channels (a byte per bit).
struct EmitJob {
    uint32_t first_op, n_ops, first_id, n_ids;
    uint64_t seq_off, contig_off;          // the bin's bases and contig bytes in the batch input ...
    uint32_t seq_bytes, contig_bytes;      // ... and how many there are
    uint32_t sig_len, begin_cut, end_cut, pad;
    uint32_t cap[ECH_COUNT + 1];           // room of every stream in its units (bytes, pairs, bits; [ECH_COUNT]: LzId bytes)
    uint32_t item[ECH_COUNT + 1];          // stream item per channel; [ECH_COUNT] = LzId; 0xFFFFFFFF: the bin has no such stream (single-end)
    uint64_t out_off[ECH_COUNT + 1];       // where the stream's bytes go
    uint64_t raw_off[ECH_COUNT];           // bit channels only (ECH_MATCH_BITS, ECH_CMATCH_BITS, ECH_MATCH_BITS_PE)
    uint8_t dna_to_idx[128];               // character -> the archive's symbol index (MinimizerParameters::dnaSymbolOrder; 255: none), as the host's d2i
};
struct EmitPlan { uint64_t jobs_off = 0, ops_off = 0, ids_off = 0; uint32_t n_jobs = 0, n_ops = 0, n_ids = 0; uint64_t out_bytes = 0; };

enum : uint32_t { MAX_STREAMS = 23 };

// per-bin block assembly plan (reference block layout: SURVEY §8 a13,
// /root/reference/fastore/fastore_pack/FastqCompressor.cpp:56-69, 684-699, 1055-1126, 1199-1210)
struct BlockPlan {
    uint64_t block_off;               // offset of the block in the compact output buffer
    uint64_t records;                 // header fields
    uint64_t raw_dna_size;
    uint64_t raw_id_size;
    uint32_t signature;
    uint32_t n_streams;               // 15 (SE) or 23 (PE)
    uint32_t first_item;              // items [first_item, first_item + n_streams) in STREAM order
    uint8_t min_len, max_len, has_headers, pad;
    uint32_t copy_order[MAX_STREAMS]; // stream indices in the order their bytes are laid out
    uint64_t work_size[MAX_STREAMS];  // pre-entropy ("work") sizes for the header
};

}  // namespace fsdev
