// Stream emission, a WAVEFRONT per op (round 5).  What an op writes is defined by emit_core.h: emit_op (the serial form: the
// test emulation's, and the oracle of this file's tests); here the same outputs come from 64 lanes that take 64 consecutive positions of
// the record at a time:
//   * every lane loads the bytes of ITS position (the read's base, the matched read's, the contig's): 64 consecutive bytes a load
//     instruction, where round 4's thread-per-op walk had every lane of a wavefront in another record (11 GB fetched and 23 GB written
//     a step for 3 GB of bases and 1 GB of streams: profiles/r04_hbm_traffic.json);
//   * "this position writes a letter / a match bit" is a ballot; a letter's place behind the op's first is the population count of the
//     ballot below the lane; the match bits of 64 positions ARE the ballot of "equal" -- they go out as bits of a packed word array
//     (LSB first) with at most three 32-bit atomic ORs a chunk, not as a byte each, and fs_rle_binary reads sixteen of them with one
//     shift (BinaryRleEncoder, rle/RleEncoder.h:21-79);
//   * the positions that write nothing -- the signature, the overlap of a shift-only match -- are lanes that stay out of the ballots.
// Two passes as before (the channels are shared by a bin's ops, so an op's place is the sum of what the ops in front of it write):
// WRITE = false counts, WRITE = true writes behind fs_emit_scan's sums.  Compiled for the device and, with -DFS_SIMT_EMU, for the
// lock-step emulation (tests/test_simt.py holds it against emit_op on every golden bin).
// Reference (fastore_pack/FastqCompressor.cpp): CompressHardRead :1388-1410, CompressNormalMatch :1460-1560, CompressContigRead
// :1690-1760, StoreContigDefinition :1620-1680, LzCompressorPE::CompressPair's letter and match loops :4790-4900.
#pragma once
#include "wave.h"
#include "emit_core.h"

namespace fsemit {

struct WaveCount { uint32_t nL, nB; };

#if defined(__HIP_DEVICE_COMPILE__)
  #define FS_BITS_OR(p, v) ((void)atomicOr((p), (v)))
#else
  #define FS_BITS_OR(p, v) (*(p) |= (v))
#endif
// (under hipcc the functions are __host__ __device__: the compiler's host pass resolves a kernel's call to a TEMPLATE strictly, and a
// host-only candidate -- what FS_DEV is in that pass -- is "not viable" there; the host instance is never called)
#if defined(__HIPCC__)
  #define FS_WAVE_FN __host__ __device__ inline
#else
  #define FS_WAVE_FN FS_DEV
#endif

// n bits (n <= 64, LSB first) behind bit `at` of the packed array: lanes 0..2 take a 32-bit word each
FS_WAVE_FN void append_bits(uint32_t* words, uint32_t at, uint64_t bits, uint32_t n)
{
    if (n == 0u) return;
    const uint32_t sh = at & 31u, lane = (uint32_t)FS_LANE();
    const uint64_t lo = bits << sh, hi = sh ? bits >> (64u - sh) : 0ull;
    const uint32_t w = lane == 0u ? (uint32_t)lo : (lane == 1u ? (uint32_t)(lo >> 32) : (uint32_t)hi);
    if (lane < 3u && w != 0u) FS_BITS_OR(words + (at >> 5) + lane, w);
}

// outL: the op's first letter / byte in its L channel; outSym: its first (match symbol, 0) pair; outBits + bitPos: its first match bit
template <bool WRITE>
FS_WAVE_FN WaveCount emit_op_wave(const fsdev::EmitOp& op, const fsdev::EmitJob& job, const uint8_t* seq, const uint8_t* contig, uint8_t* outL, uint8_t* outSym, uint32_t* outBits, uint32_t bitPos)
{
    using namespace fsdev;
    const uint32_t lane = (uint32_t)FS_LANE();
    const uint64_t below = lane ? (~0ull >> (64u - lane)) : 0ull;
    const uint8_t* d2i = job.dna_to_idx;
    const uint32_t sigLen = job.sig_len, idxN = d2i['N'];
    const uint32_t kind = op.kind, mode = op.mode;
    const bool pe = kind == EMIT_PE_MATCH, byteChannel = kind == EMIT_HARD || kind == EMIT_PE_HARD;
    const bool bitChannel = (kind == EMIT_CDEF) || ((kind == EMIT_MATCH || kind == EMIT_PE_MATCH) && mode == EMIT_FULL);
    uint32_t nL = 0, nB = 0;
    // the positions the op walks: [first, last)
    uint32_t first = 0, last = 0;
    // MATCH / PE_MATCH
    const int32_t shift = op.shift;
    const uint32_t neg = shift < 0 ? (uint32_t)(-shift) : 0u, posS = shift > 0 ? (uint32_t)shift : 0u;
    const uint32_t bestOff = op.seq_b + posS, bestLen = (uint32_t)op.len_b - posS, newLen = (uint32_t)op.len_a - neg, bestPos = (uint32_t)op.pos_b - posS;
    const uint32_t minLen = bestLen < newLen ? bestLen : newLen;
    // CREAD / CDEF
    const uint32_t readLenC = kind == EMIT_CDEF ? op.pos_b : op.len_a;
    switch (kind) {
    case EMIT_HARD: case EMIT_PE_HARD: case EMIT_MATCH: case EMIT_PE_MATCH: case EMIT_CREAD: first = 0; last = op.len_a; break;
    case EMIT_CDEF: first = op.len_a; last = op.len_b; break;
    default: break;
    }
    for (uint32_t c0 = first; c0 < last; c0 += (uint32_t)FS_WAVE) {
        const uint32_t q = c0 + lane;
        const bool in = q < last;
        bool hasL = false, hasB = false, bVal = false; uint32_t l0 = 0, l1 = 0;
        switch (kind) {
        case EMIT_HARD: {
            const uint32_t m = op.pos_a;
            if (in) { if (q < m || q >= m + sigLen) { hasL = true; l0 = seq[op.seq_a + q]; } else if (q == m) { hasL = true; l0 = '.'; } }
            break;
        }
        case EMIT_PE_HARD:
            if (in) { hasL = true; l0 = seq[op.seq_a + q]; }
            break;
        case EMIT_MATCH: case EMIT_PE_MATCH:
            if (in) {
                const uint32_t cn = seq[op.seq_a + q];
                if (q < neg) { hasL = true; l0 = d2i[cn & 127u]; l1 = idxN; }
                else {
                    const uint32_t i = q - neg;
                    if (i >= minLen) { hasL = true; l0 = d2i[cn & 127u]; l1 = idxN; }
                    else if (mode == EMIT_FULL || mode == EMIT_EXPENSIVE) {
                        const bool skipped = !pe && bestPos < minLen && i >= bestPos && i - bestPos < sigLen;      // (the signature is not coded; the mate has none)
                        if (!skipped) {
                            const uint32_t cb = seq[bestOff + i];
                            hasB = true; bVal = cb == cn;
                            if (!bVal) { hasL = true; l0 = d2i[cn & 127u]; l1 = d2i[cb & 127u]; }
                        }
                    }
                }
            }
            break;
        case EMIT_CREAD:
            if (in) {
                const uint32_t readLen = op.len_a, m = op.pos_a, consStart = readLen - m, tailAt = readLen - job.end_cut;
                const bool skipped = m < tailAt && q >= m && q - m < sigLen;
                if (!skipped) {
                    const bool always = q < job.begin_cut || q >= tailAt;
                    if (always || contig[op.seq_b + 2u * op.pos_b + consStart + q] != 0u) { hasL = true; l0 = d2i[seq[op.seq_a + q] & 127u]; l1 = d2i[contig[op.seq_b + consStart + q] & 127u]; }
                }
            }
            break;
        case EMIT_CDEF:
            if (in) {
                const uint32_t readLen = readLenC, lzFirst = readLen - op.pos_a, lzSecond = lzFirst + readLen;
                const bool skipped = first <= readLen && q >= readLen && q - readLen < sigLen;
                if (!skipped) {
                    const uint32_t v = contig[op.seq_b + 2u * readLen + q];
                    hasB = true; bVal = v == 0u;
                    if (q < lzFirst + 2u || q >= lzSecond - 2u || v != 0u) { hasL = true; l0 = d2i[contig[op.seq_b + q] & 127u]; l1 = idxN; }
                }
            }
            break;
        default: break;
        }
        const uint64_t mL = fs_ballot(hasL), mB = fs_ballot(hasB);
        if (WRITE && hasL) {
            const uint32_t at = nL + fs_popc64(mL & below);
            if (byteChannel) outL[at] = (uint8_t)l0;
            else { outL[2u * at] = (uint8_t)l0; outL[2u * at + 1u] = (uint8_t)l1; }
        }
        if (bitChannel) {
            if (WRITE) {
                // the lanes that write a bit are runs of consecutive lanes (the signature's gap is the one hole): run by run
                const uint64_t v = fs_ballot(hasB && bVal);
                uint32_t at = bitPos + nB;
                for (uint64_t left = mB; left != 0ull;) {
                    const uint32_t lo = fs_ctz64(left);
                    const uint64_t up = left >> lo;
                    const uint32_t run = ~up == 0ull ? 64u : fs_ctz64(~up);
                    const uint64_t ones = run >= 64u ? ~0ull : ((1ull << run) - 1ull);
                    append_bits(outBits, at, (v >> lo) & ones, run);
                    at += run;
                    left &= ~(ones << lo);
                }
            }
        } else if (WRITE && hasB) {
            const uint32_t at = nB + fs_popc64(mB & below);
            outSym[2u * at] = bVal ? 1u : 0u; outSym[2u * at + 1u] = 0u;
        }
        nL += fs_popc64(mL); nB += fs_popc64(mB);
    }
    WaveCount r; r.nL = nL; r.nB = nB;
    return r;
}

}  // namespace fsemit
