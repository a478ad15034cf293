// Stream emission of one record: what the ops of fsdev::EmitOp write -- the serial form: the definition the kernels'
// wavefront-per-op form (emit_wave.h) is held against, and the test-only host emulation's own.
//
// Reference (fastore_pack/FastqCompressor.cpp): CompressHardRead :1388-1410, CompressNormalMatch :1460-1560, CompressContigRead
// :1690-1760, StoreContigDefinition :1620-1680, LzCompressorPE::CompressPair's letter and match loops :4790-4900.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "device_types.h"

#if defined(__HIPCC__)
  #define FS_EMIT_FN __host__ __device__ inline
#else
  #define FS_EMIT_FN inline
#endif

namespace fsemit {

// what one op writes: L = bytes (HARD) or (symbol, context) pairs, B = match bits (a byte each) or (match symbol, 0) pairs
struct Sink {
    uint32_t nL = 0, nB = 0;
    uint8_t* outL = nullptr; uint8_t* outB = nullptr;        // null: count only
};
FS_EMIT_FN void put_byte(Sink& s, uint32_t b) { if (s.outL) s.outL[s.nL] = (uint8_t)b; s.nL++; }
FS_EMIT_FN void put_letter(Sink& s, uint32_t sym, uint32_t ctx) { if (s.outL) { s.outL[2u * s.nL] = (uint8_t)sym; s.outL[2u * s.nL + 1u] = (uint8_t)ctx; } s.nL++; }
FS_EMIT_FN void put_bit(Sink& s, bool b) { if (s.outB) s.outB[s.nB] = b ? 1u : 0u; s.nB++; }
FS_EMIT_FN void put_match_symbol(Sink& s, bool b) { if (s.outB) { s.outB[2u * s.nB] = b ? 1u : 0u; s.outB[2u * s.nB + 1u] = 0u; } s.nB++; }

// the channels an op writes to (fsdev::ECH_*; ECH_COUNT: none)
FS_EMIT_FN uint32_t channel_l(const fsdev::EmitOp& op)
{
    switch (op.kind) {
    case fsdev::EMIT_HARD: return fsdev::ECH_HARD;
    case fsdev::EMIT_MATCH: return fsdev::ECH_LETTERS;
    case fsdev::EMIT_CREAD: case fsdev::EMIT_CDEF: return fsdev::ECH_CLETTERS;
    case fsdev::EMIT_PE_HARD: return fsdev::ECH_HARD_PE;
    case fsdev::EMIT_PE_MATCH: return fsdev::ECH_LETTERS_PE;
    default: return fsdev::ECH_COUNT;
    }
}
FS_EMIT_FN uint32_t channel_b(const fsdev::EmitOp& op)
{
    switch (op.kind) {
    case fsdev::EMIT_MATCH: return op.mode == fsdev::EMIT_FULL ? fsdev::ECH_MATCH_BITS : (op.mode == fsdev::EMIT_EXPENSIVE ? fsdev::ECH_MATCH_BIN : fsdev::ECH_COUNT);
    case fsdev::EMIT_CDEF: return fsdev::ECH_CMATCH_BITS;
    case fsdev::EMIT_PE_MATCH: return op.mode == fsdev::EMIT_FULL ? fsdev::ECH_MATCH_BITS_PE : (op.mode == fsdev::EMIT_EXPENSIVE ? fsdev::ECH_MATCH_BIN_PE : fsdev::ECH_COUNT);
    default: return fsdev::ECH_COUNT;
    }
}
// bytes per unit of a channel's L / B output as the stream holds it
FS_EMIT_FN uint32_t unit_l(uint32_t ch) { return (ch == fsdev::ECH_HARD || ch == fsdev::ECH_HARD_PE) ? 1u : 2u; }
FS_EMIT_FN bool is_bit_channel(uint32_t ch) { return ch == fsdev::ECH_MATCH_BITS || ch == fsdev::ECH_CMATCH_BITS || ch == fsdev::ECH_MATCH_BITS_PE; }

// How the serial form reads the bin's bytes: a load per byte.  (The kernels take a wavefront per op: emit_wave.h.)
struct Direct {
    const uint8_t* base;
    FS_EMIT_FN explicit Direct(const uint8_t* b) : base(b) {}
    FS_EMIT_FN uint32_t get(uint32_t off) const { return base[off]; }
};

// seq: the bin's bases; contig: the bin's contig bytes (per contig: sequence[2 L] then variant[2 L], L = the contig's read length)
template <class R>
FS_EMIT_FN void emit_op_with(const fsdev::EmitOp& op, const fsdev::EmitJob& job, const uint8_t* seq, const uint8_t* contig, Sink& s)
{
    const uint8_t* d2i = job.dna_to_idx;
    const uint32_t sigLen = job.sig_len;
    const uint32_t idxN = d2i['N'];
    switch (op.kind) {
    case fsdev::EMIT_HARD: {
        R a(seq);
        const int32_t L = (int32_t)op.len_a, m = (int32_t)op.pos_a;
        for (int32_t i = 0; i < L; ++i) {
            if (i < m || i >= m + (int32_t)sigLen) put_byte(s, a.get(op.seq_a + (uint32_t)i));
            else if (i == m) put_byte(s, '.');
        }
        break;
    }
    case fsdev::EMIT_PE_HARD: {
        R a(seq);
        for (uint32_t i = 0; i < op.len_a; ++i) put_byte(s, a.get(op.seq_a + i));
        break;
    }
    case fsdev::EMIT_MATCH: case fsdev::EMIT_PE_MATCH: {
        const bool pe = op.kind == fsdev::EMIT_PE_MATCH;
        R rn(seq), rb(seq);
        uint32_t newOff = op.seq_a, bestOff = op.seq_b;
        uint32_t newLen = op.len_a, bestLen = op.len_b, bestPos = op.pos_b;
        const int32_t shift = op.shift;
        if (shift >= 0) { bestOff += (uint32_t)shift; bestLen -= (uint32_t)shift; bestPos -= (uint32_t)shift; }
        else {
            for (int32_t i = 0; i < -shift; ++i) put_letter(s, d2i[rn.get(newOff + (uint32_t)i) & 127u], idxN);
            newOff += (uint32_t)(-shift); newLen -= (uint32_t)(-shift);
        }
        const uint32_t minLen = bestLen < newLen ? bestLen : newLen;
        if (op.mode == fsdev::EMIT_FULL || op.mode == fsdev::EMIT_EXPENSIVE) {
            for (uint32_t i = 0; i < minLen; ++i) {
                if (!pe && i == bestPos) { i += sigLen - 1u; continue; }      // (the signature is not coded; the mate has none)
                const uint32_t cb = rb.get(bestOff + i), cn = rn.get(newOff + i);
                const bool eq = cb == cn;
                if (op.mode == fsdev::EMIT_FULL) put_bit(s, eq); else put_match_symbol(s, eq);
                if (!eq) put_letter(s, d2i[cn & 127u], d2i[cb & 127u]);
            }
        }
        for (uint32_t i = minLen; i < newLen; ++i) put_letter(s, d2i[rn.get(newOff + i) & 127u], idxN);
        break;
    }
    case fsdev::EMIT_CREAD: {
        R a(seq), cs(contig), var(contig);
        const uint32_t readLen = op.len_a, m = op.pos_a;
        const uint32_t csOff = op.seq_b, varOff = op.seq_b + 2u * op.pos_b;      // (pos_b: the contig's read length)
        const uint32_t consStart = readLen - m;
        uint32_t it = 0;
        while (it < job.begin_cut) {
            if (it == m) { it += sigLen; continue; }
            put_letter(s, d2i[a.get(op.seq_a + it) & 127u], d2i[cs.get(csOff + consStart + it) & 127u]); it++;
        }
        while (it < readLen - job.end_cut) {
            if (it == m) { it += sigLen; continue; }
            if (var.get(varOff + consStart + it)) put_letter(s, d2i[a.get(op.seq_a + it) & 127u], d2i[cs.get(csOff + consStart + it) & 127u]);
            it++;
        }
        while (it < readLen) { put_letter(s, d2i[a.get(op.seq_a + it) & 127u], d2i[cs.get(csOff + consStart + it) & 127u]); it++; }
        break;
    }
    case fsdev::EMIT_CDEF: {
        R cs(contig), var(contig);
        const uint32_t readLen = op.pos_b, mainSigPos = op.pos_a;
        const uint32_t csOff = op.seq_b, varOff = op.seq_b + 2u * readLen;
        const uint32_t lzFirst = readLen - mainSigPos, lzSecond = lzFirst + readLen;
        for (uint32_t i = op.len_a; i < op.len_b; ++i) {
            if (i == readLen) { i += sigLen - 1u; continue; }
            const uint32_t v = var.get(varOff + i);
            put_bit(s, v == 0u);
            if (i < lzFirst + 2u || i >= lzSecond - 2u || v != 0u) put_letter(s, d2i[cs.get(csOff + i) & 127u], idxN);
        }
        break;
    }
    default: break;
    }
}
FS_EMIT_FN void emit_op(const fsdev::EmitOp& op, const fsdev::EmitJob& job, const uint8_t* seq, const uint8_t* contig, Sink& s) { emit_op_with<Direct>(op, job, seq, contig, s); }

// ---- the run-length coders, one symbol at a time (the emulation's form, and the tail of the kernels' chunks) ----
// BinaryRleEncoder (rle/RleEncoder.h:21-79): a byte per zero -- the ones in front of it + 2 when there are any, else 0 --, a byte 255
// per 253 ones in a row, and the ones left at the end + 2.  Returns the bytes written.
FS_EMIT_FN uint32_t rle_binary_serial(const uint8_t* bits, uint32_t n, uint8_t* out)
{
    uint32_t cur = 0, w = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (bits[i]) { if (++cur == 253u) { out[w++] = 255u; cur = 0; } }
        else { out[w++] = cur ? (uint8_t)(cur + 2u) : (uint8_t)0; cur = 0; }
    }
    if (cur) out[w++] = (uint8_t)(cur + 2u);
    return w;
}
// Rle0Encoder (rle/RleEncoder.h:140-212): zeros in pairs -- a byte 0 per pair, a byte 1 for one left over in front of the next
// value or the end --, a value v > 0 as v + 1 in one byte (< 253), three (0xFE, 16 bits) or five (0xFF, 32 bits)
FS_EMIT_FN uint32_t rle0_value_bytes(uint32_t v) { const uint32_t ss = v + 1u; return ss < 253u ? 1u : (ss < 65535u ? 3u : 5u); }
FS_EMIT_FN uint32_t rle0_put_value(uint32_t v, uint8_t* o)
{
    const uint32_t ss = v + 1u;
    if (ss < 253u) { o[0] = (uint8_t)ss; return 1u; }
    if (ss < 65535u) { o[0] = 0xFE; o[1] = (uint8_t)(ss >> 8); o[2] = (uint8_t)ss; return 3u; }
    o[0] = 0xFF; o[1] = (uint8_t)(ss >> 24); o[2] = (uint8_t)(ss >> 16); o[3] = (uint8_t)(ss >> 8); o[4] = (uint8_t)ss; return 5u;
}
FS_EMIT_FN uint32_t rle0_serial(const uint32_t* v, uint32_t n, uint8_t* out)
{
    uint32_t prev = 0, w = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (v[i] == 0u) { if (prev == 0u) prev = 1u; else { out[w++] = 0u; prev = 0u; } }
        else { if (prev == 1u) { out[w++] = 1u; prev = 0u; } w += rle0_put_value(v[i], out + w); }
    }
    if (prev == 1u) out[w++] = 1u;
    return w;
}

// Host-side validation of an emission plan: every op inside its bin's bases and contig bytes, every stream inside the emission's
// region with room for the most its ops can write, every item an item.  The kernels trust their descriptors; the engine (and the
// test emulation) call this first.  Returns nullptr, or what is wrong (badJob: where).
inline const char* plan_error(const uint8_t* input, size_t inputBytes, const fsdev::EmitPlan& plan, uint32_t nItems, uint32_t& badJob)
{
    using namespace fsdev;
    badJob = 0;
    if ((plan.jobs_off & 7u) || (plan.ops_off & 7u) || (plan.ids_off & 3u) || plan.jobs_off + (uint64_t)plan.n_jobs * sizeof(EmitJob) > inputBytes ||
        plan.ops_off + (uint64_t)plan.n_ops * sizeof(EmitOp) > inputBytes || plan.ids_off + 4ull * plan.n_ids > inputBytes) return "plan outside the batch input";
    const EmitJob* jobs = (const EmitJob*)(input + plan.jobs_off); const EmitOp* ops = (const EmitOp*)(input + plan.ops_off);
    for (uint32_t j = 0; j < plan.n_jobs; ++j) {
        badJob = j;
        const EmitJob& jb = jobs[j];
        if (!((uint64_t)jb.first_op + jb.n_ops <= plan.n_ops && (uint64_t)jb.first_id + jb.n_ids <= plan.n_ids && jb.seq_off + jb.seq_bytes <= inputBytes &&
              jb.contig_off + jb.contig_bytes <= inputBytes && jb.sig_len >= 1u && jb.sig_len <= 32u)) return "ops, ids, bases or contig bytes outside the batch input";
        // (the bins' bases and contig bytes are placed on sixteen-byte boundaries)
        if ((jb.seq_off & 15u) != 0u || (jb.contig_off & 15u) != 0u || ((jb.seq_off + jb.seq_bytes + 15u) & ~15ull) > ((inputBytes + 15u) & ~15ull) || ((jb.contig_off + jb.contig_bytes + 15u) & ~15ull) > ((inputBytes + 15u) & ~15ull))
            return "bases or contig bytes not placed on sixteen-byte boundaries";
        uint64_t need[ECH_COUNT + 1] = {0};
        need[ECH_COUNT] = 6ull * jb.n_ids + 2u;
        for (uint32_t k = 0; k < jb.n_ops; ++k) {
            const EmitOp& op = ops[jb.first_op + k];
            const uint32_t chL = channel_l(op), chB = channel_b(op);
            if (op.pad2[0] != j || chL >= ECH_COUNT) return "an op of another job or of no kind";
            bool ok = true;
            switch (op.kind) {
            case EMIT_HARD: case EMIT_PE_HARD: ok = (uint64_t)op.seq_a + op.len_a <= jb.seq_bytes; need[chL] += op.len_a + (op.kind == EMIT_HARD ? 1u : 0u); break;
            case EMIT_MATCH: case EMIT_PE_MATCH: {
                const uint32_t as = (uint32_t)(op.shift < 0 ? -op.shift : op.shift);
                ok = (uint64_t)op.seq_a + op.len_a <= jb.seq_bytes && (uint64_t)op.seq_b + op.len_b <= jb.seq_bytes && as <= op.len_a && as <= op.len_b && op.mode <= EMIT_EXPENSIVE;
                need[chL] += op.len_a; if (chB < ECH_COUNT) need[chB] += op.len_a;
                break; }
            case EMIT_CREAD: ok = (uint64_t)op.seq_a + op.len_a <= jb.seq_bytes && op.pos_a <= op.len_a && op.pos_b == op.len_a && (uint64_t)op.seq_b + 4ull * op.pos_b <= jb.contig_bytes &&
                                  jb.end_cut <= op.len_a; need[chL] += op.len_a; break;
            case EMIT_CDEF: ok = op.len_a <= op.len_b && op.len_b <= 2u * op.pos_b && op.pos_a <= op.pos_b && (uint64_t)op.seq_b + 4ull * op.pos_b <= jb.contig_bytes;
                            need[chL] += 2u * op.pos_b; need[chB] += 2u * op.pos_b; break;
            default: ok = false;
            }
            if (!ok) return "an op outside its bin's bases or contig bytes";
        }
        for (uint32_t c = 0; c <= ECH_COUNT; ++c) {
            if (need[c] == 0 && jb.item[c] == 0xFFFFFFFFu) continue;
            const uint64_t unit = c == ECH_COUNT ? 1u : (is_bit_channel(c) ? 1u : ((c == ECH_HARD || c == ECH_HARD_PE) ? 1u : 2u));
            if (!(jb.item[c] < nItems && need[c] <= jb.cap[c] && jb.out_off[c] + unit * jb.cap[c] + 2u <= plan.out_bytes &&
                  (c == ECH_COUNT || !is_bit_channel(c) || jb.raw_off[c] + (uint64_t)jb.cap[c] <= plan.out_bytes))) return "a stream without an item or without room for what its ops can write";
        }
    }
    return nullptr;
}

}  // namespace fsemit
