// QVZ codebook parsing and quality symbolisation (see qvz.h for the reference map).
#include "qvz.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdexcept>
#include "qvz_core.h"
#include "device_types.h"

namespace fs {
namespace {

// alphabet_t of pmf.h:35-39 with the index table of pmf.cpp:376-396
struct Alphabet {
    std::vector<uint8_t> symbols;
    uint16_t index[QVZ_INDEX_SLOTS];
    void computeIndex()
    {
        for (uint16_t& v : index) v = 0xFFFF;
        for (size_t i = 0; i < symbols.size(); ++i) {
            if (symbols[i] >= QVZ_INDEX_SLOTS) throw std::runtime_error("QVZ codebook: quantizer output outside the quality alphabet");
            index[symbols[i]] = (uint16_t)i;
        }
    }
};

// find_output_alphabet (quantizer.cpp:449-480): the run heads of the table up to the first unused entry
Alphabet outputAlphabet(const uint8_t* q)
{
    Alphabet a;
    uint8_t p = q[0];
    if (p != QVZ_NOT_SYMBOL) a.symbols.push_back(p);
    for (uint32_t x = 1; x < QVZ_ALPHABET; ++x) {
        if (q[x] != p) {
            p = q[x];
            if (p == QVZ_NOT_SYMBOL) break;
            a.symbols.push_back(p);
        }
    }
    a.computeIndex();
    return a;
}

// alphabet_union (pmf.cpp:322-367): merge of two sorted symbol lists
Alphabet alphabetUnion(const Alphabet& a, const Alphabet& b)
{
    Alphabet r;
    size_t i = 0, j = 0;
    while (i < a.symbols.size() && j < b.symbols.size()) {
        if (a.symbols[i] < b.symbols[j]) r.symbols.push_back(a.symbols[i++]);
        else if (a.symbols[i] == b.symbols[j]) { r.symbols.push_back(a.symbols[i]); ++i; ++j; }
        else r.symbols.push_back(b.symbols[j++]);
    }
    while (i < a.symbols.size()) r.symbols.push_back(a.symbols[i++]);
    while (j < b.symbols.size()) r.symbols.push_back(b.symbols[j++]);
    r.computeIndex();
    return r;
}

}  // namespace

void QvzModel::parse(BitReader& r)
{
    const uint64_t start = r.position();
    r.getBytes(wellSeed, sizeof wellSeed);
    r.getBytes(&maxReadLength, 4);
    if (maxReadLength == 0 || maxReadLength >= 255) throw std::runtime_error("QVZ footer: bad max read length");
    const uint32_t columns = maxReadLength;
    colCtxBase.assign(columns, 0); colIndex.assign((size_t)columns * QVZ_INDEX_SLOTS, 0xFFFF);
    qratio.clear(); quant.clear(); stateOf.clear(); card.clear();
    nCtx = 0;

    auto setQuantizer = [&](uint32_t ctx, const uint8_t* line) -> Alphabet {    // COPY_Q_FROM_LINE: stored value - 33 (mod 256)
        uint8_t* q = &quant[(size_t)ctx * QVZ_ALPHABET]; uint8_t* st = &stateOf[(size_t)ctx * QVZ_ALPHABET];
        for (uint32_t i = 0; i < QVZ_ALPHABET; ++i) q[i] = (uint8_t)(line[i] - 33);
        Alphabet out = outputAlphabet(q);
        if (out.symbols.empty() || out.symbols.size() > fsqvz::MAX_CARD) throw std::runtime_error("QVZ codebook: empty quantizer");
        for (uint32_t i = 0; i < QVZ_ALPHABET; ++i) st[i] = (q[i] < QVZ_INDEX_SLOTS && out.index[q[i]] != 0xFFFF) ? (uint8_t)out.index[q[i]] : 0xFF;
        card[ctx] = (uint8_t)out.symbols.size();
        return out;
    };
    auto grow = [&](uint32_t n) { quant.resize((size_t)n * QVZ_ALPHABET); stateOf.resize((size_t)n * QVZ_ALPHABET); card.resize(n); };

    uint8_t line[QVZ_ALPHABET];
    // column 0: a single left context (symbol 0)
    Alphabet in0; in0.symbols.push_back(0); in0.computeIndex();
    memcpy(&colIndex[0], in0.index, sizeof in0.index);
    qratio.push_back((uint8_t)(r.getByte() - 33));
    if (r.get2Bytes() != QVZ_ALPHABET) throw std::runtime_error("QVZ codebook: unexpected alphabet size");
    grow(2);
    r.getBytes(line, QVZ_ALPHABET); Alphabet lo = setQuantizer(0, line);
    r.getBytes(line, QVZ_ALPHABET); Alphabet hi = setQuantizer(1, line);
    Alphabet uniques = alphabetUnion(lo, hi);
    nCtx = 2;

    std::vector<uint8_t> ratios;
    for (uint32_t c = 1; c < columns; ++c) {
        const uint32_t size = (uint32_t)uniques.symbols.size();
        colCtxBase[c] = nCtx;
        memcpy(&colIndex[(size_t)c * QVZ_INDEX_SLOTS], uniques.index, sizeof uniques.index);
        const uint32_t part = r.get2Bytes();
        if (part < size) throw std::runtime_error("QVZ codebook: ratio line shorter than the column's alphabet");
        ratios.resize(part); r.getBytes(ratios.data(), part);
        for (uint32_t i = 0; i < size; ++i) qratio.push_back((uint8_t)(ratios[i] - 33));
        // contexts are interleaved (2*idx low, 2*idx+1 high) while the file holds all lows, then all highs;
        // the next column's input alphabet is accumulated in file order (QVZ.cpp:277-297)
        grow(nCtx + 2 * size);
        Alphabet next; next.computeIndex();
        for (uint32_t pass = 0; pass < 2; ++pass)
            for (uint32_t i = 0; i < size; ++i) {
                r.getBytes(line, QVZ_ALPHABET);
                next = alphabetUnion(next, setQuantizer(nCtx + 2 * i + pass, line));
            }
        nCtx += 2 * size;
        uniques = next;
    }
    if (nCtx >= (1u << 24)) throw std::runtime_error("QVZ codebook: too many contexts");
    const uint64_t end = r.position();
    footerBytes.assign(r.data() + start, r.data() + end);            // WriteCodebook(ReadCodebook(x)) == x: kept verbatim for .cmeta
    present = true;

    // ---- device blob ----
    std::vector<fsqvz::Desc> desc(nCtx);
    uint32_t words = 0;
    for (uint32_t i = 0; i < nCtx; ++i) { desc[i].off = words; desc[i].card = card[i]; words += 1u + card[i]; }
    if (getenv("FS_TRACE")) { uint32_t mx = 0; for (uint32_t i = 0; i < nCtx; ++i) mx = std::max<uint32_t>(mx, card[i]); fprintf(stderr, "[trace] QVZ codebook: %u columns, %u contexts, %u words of counts, %.1f symbols a context, at most %u\n", columns, nCtx, words, nCtx ? (double)(words - nCtx) / nCtx : 0.0, mx); }
    fsqvz::ModelHeader h{nCtx, words, columns, 0};
    blob.assign(fsqvz::blob_bytes(nCtx, words), 0);
    memcpy(blob.data(), &h, sizeof h);
    memcpy(blob.data() + sizeof h, desc.data(), nCtx * sizeof(fsqvz::Desc));
    uint32_t* image = (uint32_t*)(blob.data() + sizeof h + nCtx * sizeof(fsqvz::Desc));
    for (uint32_t i = 0; i < nCtx; ++i) {                                // initialize_stream_stats (qv_stream.cpp:46-71)
        image[desc[i].off] = card[i];
        for (uint32_t k = 0; k < card[i]; ++k) image[desc[i].off + 1 + k] = 1;
    }

    // ---- the quantizer tables for the device's symbolisation (fs_gather_quality_qvz) ----
    {
        fsdev::QvzSymHeader sh; memset(&sh, 0, sizeof sh);
        auto place = [](size_t& at, size_t bytes) { const size_t o = at; at = (at + bytes + 15u) & ~(size_t)15u; return (uint32_t)o; };
        size_t at = (sizeof sh + 15u) & ~(size_t)15u;
        sh.columns = columns; sh.n_ctx = nCtx;
        sh.col_ctx_base_off = place(at, 4ull * columns);
        sh.col_index_off = place(at, 2ull * columns * QVZ_INDEX_SLOTS);
        sh.qratio_off = place(at, qratio.size());
        sh.quant_off = place(at, quant.size());
        sh.state_of_off = place(at, stateOf.size());
        sh.total_bytes = (uint32_t)at; sh.well_off = (uint32_t)at; sh.well_words = 0;
        symBlob.assign(at, 0);
        memcpy(symBlob.data(), &sh, sizeof sh);
        memcpy(symBlob.data() + sh.col_ctx_base_off, colCtxBase.data(), 4ull * columns);
        memcpy(symBlob.data() + sh.col_index_off, colIndex.data(), 2ull * columns * QVZ_INDEX_SLOTS);
        memcpy(symBlob.data() + sh.qratio_off, qratio.data(), qratio.size());
        memcpy(symBlob.data() + sh.quant_off, quant.data(), quant.size());
        memcpy(symBlob.data() + sh.state_of_off, stateOf.data(), stateOf.size());
    }
    well_ = std::make_shared<WellCache>();
    well_->rng.reset(wellSeed);
}

void QvzModel::writeSymBlob(uint8_t* dst, uint32_t wellWords) const
{
    memcpy(dst, symBlob.data(), symBlob.size());
    fsdev::QvzSymHeader sh; memcpy(&sh, dst, sizeof sh);
    sh.well_off = (uint32_t)((symBlob.size() + 15u) & ~(size_t)15u); sh.well_words = wellWords; sh.total_bytes = sh.well_off + 4u * wellWords;
    memcpy(dst, &sh, sizeof sh);
    if (!well_) throw std::runtime_error("QVZ model without its generator");
    std::lock_guard<std::mutex> lk(well_->mx);
    while (well_->words.size() < wellWords) well_->words.push_back(well_->rng.next());      // (the front of well_1024a_bits: one output per four draws)
    memcpy(dst + sh.well_off, well_->words.data(), 4ull * wellWords);
}

void qvzSymbolise(const QvzModel& m, WellRng& rng, const uint8_t* qua, uint32_t len, uint32_t qualityOffset, bool reverse, std::vector<uint8_t>& out)
{
    if (len > m.maxReadLength) throw std::runtime_error("QVZ: read longer than the codebook");
    const size_t o = out.size();
    out.resize(o + (size_t)len * 4);
    uint8_t* d = out.data() + o;
    uint32_t prev = 0;
    for (uint32_t i = 0; i < len; ++i) {
        const uint32_t ii = reverse ? len - 1 - i : i;
        const uint32_t qv = (uint32_t)(uint8_t)(qua[ii] - qualityOffset);
        if (qv >= QVZ_ALPHABET) throw std::runtime_error("QVZ: quality value outside the alphabet");
        const uint32_t idx = prev < QVZ_INDEX_SLOTS ? m.colIndex[(size_t)i * QVZ_INDEX_SLOTS + prev] : 0xFFFFu;
        if (idx == 0xFFFFu) throw std::runtime_error("QVZ: previous value not in the column's input alphabet");
        const uint32_t pair = m.colCtxBase[i] / 2 + idx;                 // contexts come in low/high pairs
        const uint32_t ctx = 2 * pair + (rng.bits(7) >= m.qratio[pair] ? 1u : 0u);
        const uint32_t st = m.stateOf[(size_t)ctx * QVZ_ALPHABET + qv];
        if (st == 0xFF) throw std::runtime_error("QVZ: quantizer maps the value to an unused entry");
        const uint32_t w = ctx | (st << 24);
        memcpy(d + (size_t)i * 4, &w, 4);
        prev = m.quant[(size_t)ctx * QVZ_ALPHABET + qv];
    }
}

}  // namespace fs
