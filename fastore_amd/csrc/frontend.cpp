#include "frontend.h"
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <string.h>
#include <time.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <stdexcept>
#include <string>
#include "introsort.h"

namespace fs {

namespace {

enum NodeType : uint8_t { TYPE_NONE = 0, TYPE_HARD, TYPE_LZ, TYPE_CONTIG_READ };
enum ReadFlags { ReadIdentical = 0, ReadDifficult, ReadShiftOnly, ReadFullEncode, ReadFullExpensive, ReadTreeGroupStart,
                 ReadContigGroupStart, ReadContigGroupNext, ReadGroupEnd };
enum ReadMatchType { HardRead = 0, ExactMatch, LzMatch, ContigRead };
enum ReadFlagsPE { ReadDifficultPE = 0, ReadIdenticalPE, ReadFullEncodePE, ReadFullExpensivePE, ReadShiftOnlyPE };
const int32_t ShiftOffset = 128 + 1;

// A record as the matcher sees it.  Real records map 1:1; sub-tree roots are *copies* of a
// record with another signature position (FastqCompressor.cpp:1791-1798) and get their own id so
// that "same FastqRecord object" tests (LZ history lookup) keep their meaning.
struct VRec { uint32_t rec; uint16_t minimPos; };

struct Contig {
    std::vector<char> sequence;
    std::vector<uint8_t> variant;
    uint32_t rangeFirst = 0, rangeSecond = 0, variantsCount = 0, readLen = 0;
    std::vector<int32_t> nodes;       // contig member nodes in encode order
};

struct Node {
    int32_t vrec = -1;
    uint8_t type = TYPE_NONE;
    bool noMismatches = false, hasEm = false;
    int16_t shift = 0, cost = 0;
    int32_t lzVrec = -1, parent = -1, contig = -1;
    std::vector<int32_t> children;
    std::vector<int32_t> em;          // exact-match group: vrec ids
    std::vector<uint32_t> trees;      // sub-tree group ids (Batch::trees)
};

struct BinaryRle {                   // rle/RleEncoder.h:21-79
    std::vector<uint8_t>* w = nullptr; uint32_t cur = 0;
    void start(std::vector<uint8_t>* o) { w = o; cur = 0; }
    void put(bool s)
    {
        if (s) { cur++; if (cur == 255 - 2) { w->push_back((uint8_t)(cur + 2)); cur = 0; } }
        else {
            const bool mism = (cur > 0) && (cur < 255 - 2);
            if (cur > 0) { w->push_back((uint8_t)(cur + 2)); cur = 0; }
            if (!mism) w->push_back(0);
        }
    }
    void end() { if (cur > 0) { w->push_back((uint8_t)(cur + 2)); cur = 0; } }
};
struct Rle0 {                        // rle/RleEncoder.h:140-212
    std::vector<uint8_t>* w = nullptr; uint32_t prev = 0;
    void start(std::vector<uint8_t>* o) { w = o; prev = 0; }
    void put(uint32_t s)
    {
        if (s == 0) { if (prev == 0) prev = 1; else if (prev == 1) { w->push_back(0); prev = 0; } }
        else {
            if (prev == 1) { w->push_back(1); prev = 0; }
            const uint32_t ss = s + 1;
            if (ss < 253) w->push_back((uint8_t)ss);
            else if (ss < (1u << 16) - 1) { w->push_back(0xFE); w->push_back((uint8_t)(ss >> 8)); w->push_back((uint8_t)ss); }
            else { w->push_back(0xFF); w->push_back((uint8_t)(ss >> 24)); w->push_back((uint8_t)(ss >> 16)); w->push_back((uint8_t)(ss >> 8)); w->push_back((uint8_t)ss); }
        }
    }
    void end() { if (prev == 1) w->push_back(1); }
};

}  // namespace

uint32_t intLog(uint64_t x, uint64_t base)
{
    uint32_t r = 0;
    if (base == 0) return 1;
    if (base == 1) base++;
    // (a range with bit 31 set reaches here sign-extended: the reference's loop would then wrap its power to zero and never
    // end; seven is the last exponent whose power of 256 fits, and the byte count it stands for is the most a u64 has)
    for (uint64_t t = base; t <= x; t *= base) { ++r; if (t > ~0ull / base) break; }
    return r > 7 && base == 256 ? 7 : r;
}

bool streamIsRangeCoded(uint32_t s, uint32_t qm)
{
    switch (s) {
    case S_Rev: case S_MatchBinary: case S_LettersX: case S_CLetters: case S_IdToken: case S_IdValue:
    case S_LettersXPE: case S_MatchBinaryPE: case S_FlagPE: return true;
    case S_Quality: return qm != MET_NONE;
    default: return false;
    }
}
uint32_t streamModel(uint32_t s, uint32_t qm)
{
    // fsrc::Model ids: 0 s2o4, 1 s8o4, 2 a8o4, 3 a2o10, 4 a8o6, 5 a256o1
    switch (s) {
    case S_Rev: case S_MatchBinary: case S_MatchBinaryPE: return 0;
    case S_LettersX: case S_CLetters: case S_LettersXPE: case S_FlagPE: return 2;
    case S_IdToken: case S_IdValue: return 5;
    case S_Quality: return qm == MET_BINARY ? 3 : 4;
    default: return 0;
    }
}

void compressReadId(const HeaderStats& head, const uint8_t* h, uint32_t headLen, std::vector<uint8_t>& tok, std::vector<uint8_t>& val)
{
    uint32_t fieldStart = 0, fi = 0;
    const uint32_t nf = (uint32_t)head.fields.size();
    for (uint32_t i = 0; i <= headLen; ++i) {
        if (fi >= nf) break;                                   // more separators than fields: the reference would run off the table
        const HeaderField& f = head.fields[fi];
        if (i != headLen && (char)h[i] != f.separator) continue;
        if (f.isConst) { fieldStart = i + 1; fi++; continue; }
        const uint8_t* field = h + fieldStart;
        const uint32_t fieldLen = i - fieldStart;
        if (!f.isNumeric) {
            const std::string s((const char*)field, fieldLen);
            const auto it = std::find(f.possibleValues.begin(), f.possibleValues.end(), s);
            const uint32_t id = (uint32_t)(it - f.possibleValues.begin());
            tok.push_back((uint8_t)id); tok.push_back((uint8_t)fi);
        } else {
            uint64_t v = 0;                                     // is_num(): digits prefix
            for (uint32_t k = 0; k < fieldLen; ++k) { if (field[k] < '0' || field[k] > '9') break; v = v * 10 + (field[k] - '0'); }
            const int64_t diff = (int64_t)(v - f.minValue);
            uint32_t ctxBase = fi << 2;
            const int32_t valueRange = (int32_t)(f.maxValue - f.minValue);
            int32_t plog = (int32_t)intLog((uint64_t)(int64_t)valueRange, 256);
            while (plog >= 0) { val.push_back((uint8_t)((diff >> (8 * plog--)) & 0xFF)); val.push_back((uint8_t)ctxBase++); }
        }
        fieldStart = i + 1; fi++;
    }
}

void compressReadQuality(const BinModuleConfigRaw& cfg, const uint8_t* seq, const uint8_t* qua, uint32_t len, bool reverse, std::vector<uint8_t>& out,
                         const QvzModel* qvz, WellRng* rng)
{
    static const uint8_t q8[64] = {0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5,
                                   5, 5, 5, 6, 6, 6, 6, 6, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7};
    const uint32_t off = cfg.archiveType.qualityOffset;
    switch (cfg.quaParams.method) {
    case MET_NONE: {
        const size_t o = out.size();
        out.resize(o + len);
        uint8_t* d = out.data() + o;
        if (!reverse) for (uint32_t i = 0; i < len; ++i) d[i] = (uint8_t)(qua[i] - off);
        else for (uint32_t i = 0; i < len; ++i) d[i] = (uint8_t)(qua[len - 1 - i] - off);
        break;
    }
    case MET_BINARY:
        for (uint32_t i = 0; i < len; ++i) {
            const uint32_t ii = reverse ? len - 1 - i : i;
            if (seq[ii] == 'N') continue;
            const uint32_t q = (uint32_t)(qua[ii] - off) >= cfg.quaParams.binaryThreshold;
            out.push_back((uint8_t)q); out.push_back((uint8_t)((i * 2) / len));
        }
        break;
    case MET_8BIN:
        for (uint32_t i = 0; i < len; ++i) {
            const uint32_t ii = reverse ? len - 1 - i : i;
            if (seq[ii] == 'N') continue;
            out.push_back(q8[(qua[ii] - off) & 63]); out.push_back((uint8_t)((i * 8) / len));
        }
        break;
    case MET_QVZ:
        if (!qvz || !qvz->present || !rng) throw std::runtime_error("QVZ quality coding needs the library's codebook");
        qvzSymbolise(*qvz, *rng, qua, len, off, reverse, out);
        break;
    default: throw std::runtime_error("unknown quality coding method");
    }
}

// =================================================================================================
struct BinEncoder::Impl {
    BinModuleConfigRaw cfg{};
    const HeaderStats* headp = nullptr;
    const QvzModel* qvzp = nullptr; WellRng well;
    const PackParams par;
    uint32_t sigLen = 8;
    bool pe = false, hasHeaders = false;
    int8_t dnaToIdx[128];

    const Batch* B = nullptr;         // bases, qualities, headers, records
    const Batch* G = nullptr;         // stored graph tables (nodes, top nodes, exact-match records, sub-trees); may be B
    BinStreams* out = nullptr;
    uint32_t recBase = 0, curSig = 0;
    std::vector<VRec> vrecs;
    std::vector<Node> nodes;          // all nodes of the bin (top-level first, sub-tree nodes by Batch order)
    std::deque<Contig> contigs;       // deque: references stay valid while nested sub-trees add contigs
    uint32_t nodeBase = 0;            // Batch node index of nodes[0]

    BinaryRle matchRle, consMatchRle, matchRlePE;
    Rle0 lzRle0;

    struct LzContext { std::vector<int32_t> history; };
    std::vector<LzContext> lzStack;
    struct ConsEnc { int32_t lastMinimPos = 0; const Contig* def = nullptr; uint32_t defOff = 0; };      // defOff: the contig's bytes in out->contigBytes (device-side emission)
    // device-side emission: the walk leaves an op where it would compare bases (emit_core.h does that on the device)
    bool devEmit = false;
    uint32_t seqRel(const uint8_t* p) const { return (uint32_t)((uint64_t)(p - B->seq.data()) - out->emitSeqLo); }
    void pushOp(uint32_t kind, uint32_t a, uint32_t b, uint32_t lenA, uint32_t lenB, uint32_t posA, uint32_t posB, int32_t shift, uint32_t mode)
    {
        fsdev::EmitOp op; memset(&op, 0, sizeof op);
        op.kind = kind; op.seq_a = a; op.seq_b = b; op.len_a = (uint16_t)lenA; op.len_b = (uint16_t)lenB; op.pos_a = (uint16_t)posA; op.pos_b = (uint16_t)posB; op.shift = (int16_t)shift; op.mode = (uint8_t)mode;
        out->emitOps.push_back(op);
    }
    std::vector<ConsEnc> consStack;

    explicit Impl(const PackParams& p) : par(p) { memset(dnaToIdx, -1, sizeof dnaToIdx); }
    void setArchive(const ArchiveParams& a)
    {
        const bool sameSig = a.cfg.minimizer.signatureLen == cfg.minimizer.signatureLen &&
                             a.cfg.minimizer.signatureMaskCutoffBits == cfg.minimizer.signatureMaskCutoffBits;
        cfg = a.cfg; headp = &a.head; qvzp = &a.qvz;
        sigLen = cfg.minimizer.signatureLen; pe = cfg.archiveType.readType == READ_PE; hasHeaders = cfg.archiveType.readsHaveHeaders != 0;
        memset(dnaToIdx, -1, sizeof dnaToIdx);
        for (int i = 0; i < 5; ++i) dnaToIdx[(int)cfg.minimizer.dnaSymbolOrder[i]] = (int8_t)i;
        if (!sameSig) dropPairState();
    }
    void dropPairState();

    // ---- record accessors ----
    const Rec& R(int32_t v) const { return B->recs[vrecs[v].rec]; }
    const uint8_t* seq(int32_t v) const { return B->seq.data() + R(v).seqOff; }
    const uint8_t* qua(int32_t v) const { return B->qua.data() + R(v).seqOff; }
    uint32_t seqLen(int32_t v) const { return R(v).seqLen; }
    uint32_t minimPos(int32_t v) const { return vrecs[v].minimPos; }
    bool isReverse(int32_t v) const { return (R(v).flags & FLAG_REVERSE) != 0; }

    void putByte(uint32_t s, uint32_t b) { out->s[s].push_back((uint8_t)b); }
    void putSym(uint32_t s, uint32_t sym, uint32_t ctx = 0) { out->s[s].push_back((uint8_t)sym); out->s[s].push_back((uint8_t)ctx); }
    uint32_t d2i(uint8_t c) const { return (uint32_t)(uint8_t)dnaToIdx[c & 127]; }

    // ------------------------------------------------------------------------------------------
    // TFastqComparator / IFastqComparator::CompareReads  (fastore_bin/FastqRecord.h:226-257)
    bool compareReads(int32_t a, int32_t b) const
    {
        const Rec &r1 = R(a), &r2 = R(b);
        const uint32_t m1 = minimPos(a), m2 = minimPos(b);
        const uint8_t *p1 = seq(a) + m1, *p2 = seq(b) + m2;
        const uint32_t l1 = (uint32_t)r1.seqLen + r1.auxLen - m1, l2 = (uint32_t)r2.seqLen + r2.auxLen - m2;
        const uint32_t len = l1 < l2 ? l1 : l2;
        const int r = memcmp(p1, p2, len);               // strncmp on NUL-free bases
        if (r == 0) {
            if (m1 == m2) {
                const uint8_t *s1 = seq(a), *s2 = seq(b);
                for (int32_t i = (int32_t)m1; i >= 0; i--) { if (s1[i] < s2[i]) return true; if (s1[i] > s2[i]) return false; }
                return false;
            }
            return m1 > m2;
        }
        return r < 0;
    }

    // ------------------------------------------------------------------------------------------
    // ReadsClassifierSE::UpdateLzMatchResult (fastore_pack/ReadsClassifier.h:160-196)
    struct MatchResult { int32_t cost = 255; bool noMismatches = false; int32_t prevId = 0, shift = 0; };
    bool updateLzMatch(MatchResult& res, const uint8_t* s, uint32_t sLen, int32_t minPos, const uint8_t* lz, uint32_t lzLen, int32_t lzMinPos) const
    {
        const int32_t shift = lzMinPos - minPos;
        const int32_t ashift = shift < 0 ? -shift : shift;
        const int32_t insertCost = ashift * par.shiftCost;
        if (insertCost > res.cost || (uint32_t)ashift > 127u) return false;
        const int32_t recOff = shift < 0 ? -shift : 0, lzOff = shift > 0 ? shift : 0;
        const uint8_t *s1 = s + recOff, *s2 = lz + lzOff;
        const uint32_t a = sLen - recOff, b = lzLen - lzOff, minLen = a < b ? a : b;
        int32_t cc = insertCost;
        uint32_t i = 0;
#if defined(__SSE2__)
        // 16 bases per step.  The scan may run past the base at which the cost first reaches res.cost, but the cost only
        // grows, so the outcome (rejected) is the same; below res.cost the count is exact.
        for (; i + 16 <= minLen && cc < res.cost; i += 16) {
            const __m128i a = _mm_loadu_si128((const __m128i*)(s1 + i)), b = _mm_loadu_si128((const __m128i*)(s2 + i));
            const uint32_t eq = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(a, b));
            cc += (16 - (int32_t)__builtin_popcount(eq)) * par.mismatchCost;
        }
#endif
        for (; i < minLen && cc < res.cost; ++i) cc += (s1[i] != s2[i]) * par.mismatchCost;
        if (cc < res.cost) { res.cost = cc; res.noMismatches = (cc - insertCost == 0); res.shift = shift; return true; }
        return false;
    }

    // One of the 25 prefix buffers (the reference's std::set<MatchNode*, prefixFun>, ReadsClassifier.cpp:115-145): node ids in prefix order, in CHUNKS of at
    // most 1 024 -- an insert moves half a chunk at most.  (As one sorted vector a bin of n reads paid n^2 / 50 moves: 0.4 s of the 1.1 s its top-level tree
    // took in a 197 k-pair bin, and four times that at twice the size.)  A search looks at up to W/2+1 neighbours either way: they are copied out, in order.
    struct PrefixSet {
        enum : size_t { kMax = 1024 };
        std::vector<std::vector<int32_t>> chunks;
        struct Pos { size_t c, i; };                         // (chunks.size(), 0) = end
        bool atEnd(Pos p) const { return p.c >= chunks.size(); }
        int32_t at(Pos p) const { return chunks[p.c][p.i]; }
        template <class Less> Pos lowerBound(int32_t x, Less less) const
        {
            if (chunks.empty()) return Pos{0, 0};
            // the last chunk whose first element is less than x may hold the bound; if none is, the bound is the very first element
            size_t lo = 0, hi = chunks.size();
            while (lo < hi) { const size_t mid = (lo + hi) / 2; if (less(chunks[mid].front(), x)) lo = mid + 1; else hi = mid; }
            if (lo == 0) return Pos{0, 0};
            const std::vector<int32_t>& ch = chunks[lo - 1];
            const size_t i = (size_t)(std::lower_bound(ch.begin(), ch.end(), x, less) - ch.begin());
            if (i < ch.size()) return Pos{lo - 1, i};
            return lo < chunks.size() ? Pos{lo, 0} : Pos{chunks.size(), 0};
        }
        void forward(Pos p, size_t maxCnt, std::vector<int32_t>& out) const      // the elements from p on
        {
            out.clear();
            for (size_t c = p.c, i = p.i; c < chunks.size() && out.size() < maxCnt; ++c, i = 0) {
                const size_t take = std::min(maxCnt - out.size(), chunks[c].size() - i);
                out.insert(out.end(), chunks[c].begin() + (ptrdiff_t)i, chunks[c].begin() + (ptrdiff_t)(i + take));
            }
        }
        void backward(Pos p, size_t maxCnt, std::vector<int32_t>& out) const     // the elements in front of p, nearest first
        {
            out.clear();
            size_t c = p.c, i = p.i;
            while (out.size() < maxCnt) {
                if (i == 0) { if (c == 0) break; --c; i = chunks[c].size(); continue; }
                out.push_back(chunks[c][--i]);
            }
        }
        void insert(Pos p, int32_t x)
        {
            if (chunks.empty()) { chunks.emplace_back(); chunks[0].reserve(64); chunks[0].push_back(x); return; }
            if (atEnd(p)) { p.c = chunks.size() - 1; p.i = chunks[p.c].size(); }
            else if (p.i == 0 && p.c > 0 && chunks[p.c - 1].size() < chunks[p.c].size()) { --p.c; p.i = chunks[p.c].size(); }      // (in front of a chunk = behind the one before it)
            std::vector<int32_t>& ch = chunks[p.c];
            ch.insert(ch.begin() + (ptrdiff_t)p.i, x);
            if (ch.size() > kMax) {
                std::vector<int32_t> upper(ch.begin() + (ptrdiff_t)(kMax / 2), ch.end());
                ch.resize(kMax / 2);
                chunks.insert(chunks.begin() + (ptrdiff_t)p.c + 1, std::move(upper));
            }
        }
    };
    std::vector<int32_t> prefFwd, prefRev;                   // a search's neighbours, copied out of the set

    struct WinEntry { const uint8_t* seq; int32_t node; uint16_t seqLen, minPos; };
    // the LZ window, newest entry first: a power-of-two ring (the scan over it is the hottest loop of the front end; a
    // std::deque pays a block lookup per index)
    struct WinRing {
        std::vector<WinEntry> buf; std::vector<int16_t> mp;      // mp: the entries' minPos again, packed for the vector pre-filter
        uint32_t head = 0, count = 0, mask = 0;
        void reset(uint32_t capacity) { uint32_t c = 1; while (c < capacity + 2) c <<= 1; if (buf.size() != c) { buf.resize(c); mp.assign(c + 8, 0); } mask = c - 1; head = 0; count = 0; }
        uint32_t size() const { return count; }
        const WinEntry& operator[](uint32_t i) const { return buf[(head + i) & mask]; }
        void push_front(const WinEntry& e) { head = (head - 1) & mask; buf[head] = e; mp[head] = (int16_t)e.minPos; ++count; }
        void pop_back() { --count; }
    };
    // ---- device-side window search (matcher.hip) ----
    // Every match-tree construction of the bin is known once the bin is unpacked and its top-level reads are sorted: the
    // top-level one, and one per stored sub-tree (its reads in stored order behind a copy of the record that holds the
    // tree, FastqCompressor.cpp:1784-1818).  mReads/mCalls are that table; mRows the answers (from the device, or traced
    // from the host scan for the parity check); callOfTree[t] = construction of sub-tree t (-1 top level = call 0).
    std::vector<fsdev::MatchRead> mReads; std::vector<fsdev::MatchCall> mCalls; std::vector<fsdev::MatchRow> mRows;
    std::vector<fsdev::PackedRead> mPacked;    // the reads of mReads as the bin file stores them (bins whose .bdna bytes were kept)
    bool packedBases = false;
    const BinIn* curBin = nullptr;
    fsdev::PackedDna packedDnaBuf{};
    const fsdev::PackedDna* packedDna()
    {
        if (!packedBases || mPacked.size() != mReads.size()) return nullptr;
        packedDnaBuf.dna = B->dnaPacked.data() + curBin->dnaPackedOff; packedDnaBuf.bytes = (size_t)curBin->dnaPackedBytes; packedDnaBuf.reads = mPacked.data();
        memset(packedDnaBuf.symbol_order, 0, sizeof packedDnaBuf.symbol_order); memcpy(packedDnaBuf.symbol_order, cfg.minimizer.dnaSymbolOrder, 5);
        packedDnaBuf.sig_len = cfg.minimizer.signatureLen;
        return &packedDnaBuf;
    }
    std::vector<uint32_t> mWarm;              // warm-up lists of the pieces of the top-level construction
    struct Slot3 { uint64_t hash; uint32_t read, push; };
    std::vector<Slot3> dupTable;
    std::vector<int32_t> callOfTree;
    uint32_t topCalls = 0;                     // the first topCalls entries of mCalls are the pieces of the top-level construction
    bool havePre = false;                      // mRows hold the device's answers: constructMatchTree does not scan
    bool traceResize = false;
    std::vector<fsdev::MatchRow>* matchTrace = nullptr;   // host scan: note every read's answer here (same indexing as mRows)
    bool wantDevEmit = false;            // BinEncoder::setDeviceEmit
    AsyncMateFn asyncMates;              // BinEncoder::setAsyncMates
    // what a prefix-buffer neighbour is priced with, ONE load away from its node id (through the node, its record and the record's place in the
    // batch it is three, each a cache miss in a bin of tens of thousands of reads: a search prices up to 1 026 neighbours -- round 5: 70-145 k
    // clocks a search -> 30-65 k with this table and the bases of the neighbour six ahead asked for early)
    struct NodeSeq { const uint8_t* seq; uint16_t len, minPos; };
    std::vector<NodeSeq> nodeSeq;
    MatchFn matcher;
    std::function<bool()> matcherGate;   // BinEncoder::setMatcherGate: asked once a bin, before its table is made
    uint64_t matchSeqBase = 0, matchSeqBytes = 0;

    void buildMatchTable(const std::vector<int32_t>& topOrder)
    {
        mReads.clear(); mCalls.clear(); mPacked.clear(); callOfTree.assign(G->trees.size(), -1);
        packedBases = curBin && curBin->dnaPackedBytes != 0 && B->dnaBit.size() == B->recs.size() && cfg.minimizer.signatureLen * 2u <= fsdev::PACKED_SIG_BITS;
        uint64_t lo = ~0ull, hi = 0;
        auto add = [&](uint32_t rec, uint32_t minPos) {
            const Rec& r = B->recs[rec];
            lo = std::min<uint64_t>(lo, r.seqOff); hi = std::max<uint64_t>(hi, (uint64_t)r.seqOff + r.seqLen);
            mReads.push_back(fsdev::MatchRead{r.seqOff, r.seqLen, (uint16_t)minPos});
            if (packedBases) mPacked.push_back(fsdev::PackedRead{B->dnaBit[rec], B->dnaInfo[rec]});
        };
        // The top-level construction is long (most of a bin's reads).  Which of its reads are exact duplicates -- and stay
        // out of the window -- can be told without searching: a read is one exactly when an equal read (same bases, same
        // signature position) is still in the window, and equal reads never share the window, so one table look-up per
        // read (its class's latest non-duplicate, and how many non-duplicates came since) decides it.  That makes the
        // window in front of ANY read known (the last W-1 non-duplicates before it), so the construction is cut into
        // pieces that the device searches side by side, each starting from its warm-up list.
        mWarm.clear();
        for (int32_t n : topOrder) add(vrecs[nodes[n].vrec].rec, vrecs[nodes[n].vrec].minimPos);
        {
            const uint32_t n = (uint32_t)topOrder.size(), cap = par.maxLzWindowSize > 1 ? par.maxLzWindowSize - 1u : 1u;
            const uint32_t piece = 512u;
            std::vector<uint8_t> dup(n, 0);
            {
                struct Slot { uint64_t hash; uint32_t read, push; };                  // read + 1 (0 = empty); push number of the class's latest non-duplicate
                uint32_t tsz = 64; while (tsz < 2u * n) tsz <<= 1;
                dupTable.assign(tsz, Slot3{0, 0, 0});
                uint32_t pushed = 0;
                const NodeSeq* const nsq = nodeSeq.data();               // (bases, length, signature position a load away from the node id; the reads eight ahead asked for early)
                for (uint32_t i = 0; i < n; ++i) {
                    if (i + 16 < n) _mm_prefetch((const char*)&nsq[topOrder[i + 16]], _MM_HINT_T0);
                    if (i + 8 < n) { const uint8_t* pf = nsq[topOrder[i + 8]].seq; _mm_prefetch((const char*)pf, _MM_HINT_T0); _mm_prefetch((const char*)pf + 64, _MM_HINT_T0); _mm_prefetch((const char*)pf + 128, _MM_HINT_T0); }
                    const NodeSeq& na = nsq[topOrder[i]];
                    const uint8_t* sa = na.seq; const uint32_t la = na.len, ma = na.minPos;
                    uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)la << 32 | ma);
                    uint32_t k = 0;
                    for (; k + 8 <= la; k += 8) { uint64_t w; memcpy(&w, sa + k, 8); h = (h ^ w) * 0xff51afd7ed558ccdull; h ^= h >> 32; }
                    { uint64_t w = 0; memcpy(&w, sa + k, la - k); h = (h ^ w) * 0xc4ceb9fe1a85ec53ull; h ^= h >> 29; }
                    uint32_t at = (uint32_t)h & (tsz - 1u);
                    for (;; at = (at + 1u) & (tsz - 1u)) {
                        Slot3& s = dupTable[at];
                        if (s.read == 0) { s.hash = h; s.read = i + 1u; s.push = pushed++; break; }        // a new class
                        if (s.hash != h) continue;
                        const NodeSeq& nb = nsq[topOrder[s.read - 1u]];
                        if (nb.len != la || nb.minPos != ma || memcmp(nb.seq, sa, la) != 0) continue;
                        if (pushed - 1u - s.push < cap) dup[i] = 1;                                         // its class's non-duplicate is still in the window
                        else { s.read = i + 1u; s.push = pushed++; }                                        // it has left: this read takes its place
                        break;
                    }
                }
                (void)sizeof(Slot);
            }
            for (uint32_t c0 = 0; c0 < n || c0 == 0; c0 += piece) {
                fsdev::MatchCall c{c0, std::min(piece, n - c0), -1, (uint32_t)mWarm.size(), 0u, 0u};
                std::vector<uint32_t> w;
                for (uint32_t j = c0; j > 0 && w.size() < cap; --j) if (!dup[j - 1]) w.push_back(j - 1);
                c.warm_count = (uint32_t)w.size();
                mWarm.insert(mWarm.end(), w.rbegin(), w.rend());          // oldest first
                mCalls.push_back(c);
                if (n == 0) break;
            }
        }
        topCalls = (uint32_t)mCalls.size();
        for (size_t n = 0; n < nodes.size(); ++n) {
            const NodeIn& ni = G->nodes[nodeBase + n];
            for (uint32_t k = 0; k < ni.treeCount; ++k) {
                const uint32_t t = ni.treeBegin + k; const TreeIn& tree = G->trees[t];
                if (tree.nodeCount == 0 || (uint64_t)tree.nodeBegin - nodeBase + tree.nodeCount > nodes.size()) continue;
                fsdev::MatchCall c{0u, tree.nodeCount, (int32_t)mReads.size(), 0u, 0u, 0u};
                add(vrecs[nodes[n].vrec].rec, (uint32_t)tree.mainSignaturePos);        // the root copy, at the sub-tree's signature position
                c.first = (uint32_t)mReads.size();
                for (uint32_t j = 0; j < tree.nodeCount; ++j) { const Node& sn = nodes[tree.nodeBegin - nodeBase + j]; add(vrecs[sn.vrec].rec, vrecs[sn.vrec].minimPos); }
                callOfTree[t] = (int32_t)mCalls.size();
                mCalls.push_back(c);
            }
        }
        matchSeqBase = lo == ~0ull ? 0 : lo; matchSeqBytes = lo == ~0ull ? 0 : hi - lo;
        for (auto& r : mReads) r.seq_off -= (uint32_t)matchSeqBase;
    }
    fsdev::MatchParams matchParams() const { return fsdev::MatchParams{par.maxLzWindowSize, par.shiftCost, par.mismatchCost, par.encodeThreshold}; }

    std::vector<uint64_t> prefKey;            // per node: the packed start of its reversed prefix (see constructMatchTree)
    WinRing winRings[8];                      // one per nesting level of sub-trees (constructMatchTree re-enters itself)
    uint32_t winDepth = 0;

    // ReadsClassifierSE::ConstructMatchTree (fastore_pack/ReadsClassifier.cpp:95-442).
    // order: node ids in processing order; auxRoot: node id of the sub-tree root copy or -1.
    // call: index of this construction in mCalls (answers / trace rows at mCalls[call].first + position; the top-level
    // construction is call 0 and its pieces follow each other in the table), or -1
    void constructMatchTree(const std::vector<int32_t>& order, std::vector<int32_t>& roots, int32_t auxRoot, int32_t call = -1)
    {
        const bool pre = havePre && call >= 0;
        const fsdev::MatchRow* preRows = pre ? mRows.data() + mCalls[call].first : nullptr;
        fsdev::MatchRow* traceRows = (matchTrace && call >= 0) ? matchTrace->data() + mCalls[call].first : nullptr;
        const int32_t tableFirst = call >= 0 ? (int32_t)mCalls[call].first : 0, tableAux = call >= 0 ? mCalls[call].aux : -1;
        std::vector<int32_t> placeOf;              // trace: node -> its place in `order`
        if (traceRows) { placeOf.assign(nodes.size(), -1); for (size_t k = 0; k < order.size(); ++k) placeOf[order[k]] = (int32_t)k; }
        struct Dummy { uint8_t b[256]; Dummy() { memset(b, 'N', sizeof b); } };
        static const Dummy dummyEntry;                        // (initialised once, thread-safe: the encoders run side by side)
        const uint8_t* dummy = dummyEntry.b;
        roots.clear();
        const uint32_t W = par.maxLzWindowSize;
        struct DepthGuard { uint32_t& d; explicit DepthGuard(uint32_t& x) : d(x) { ++d; } ~DepthGuard() { --d; } } depthGuard(winDepth);
        if (winDepth > 8) throw std::runtime_error("sub-trees nested deeper than 8 levels");
        WinRing& win = winRings[winDepth - 1];    // real entries, newest first
        win.reset(W);
        uint32_t numDummies = W; bool dupAtBack = false;
        const bool usePrefix = par.extraReduceHardReads || par.extraReduceExpensiveLzMatches;
        PrefixSet rp[25];                         // the 25 std::set<MatchNode*, prefixFun>

        // every node that enters a prefix buffer has minimPos >= 8, so the first six bases before the signature (in
        // comparison order) are inside the compared range of any pair: packed into one word they decide most comparisons
        if (usePrefix && prefKey.size() < nodes.size()) prefKey.resize(nodes.size());
        auto makeKey = [&](int32_t x) {
            const int32_t mx = (int32_t)minimPos(nodes[x].vrec); const uint8_t* px = seq(nodes[x].vrec) + mx - 2;
            uint64_t k = 0; for (int i = 0; i < 6; ++i) k = (k << 8) | px[-i];
            prefKey[x] = k;
        };
        auto prefixLess = [&](int32_t x, int32_t y) {
            if (prefKey[x] != prefKey[y]) return prefKey[x] < prefKey[y];
            const int32_t mx = (int32_t)minimPos(nodes[x].vrec), my = (int32_t)minimPos(nodes[y].vrec);
            const int32_t maxRange = (mx < my ? mx : my) - 2;
            const uint8_t *px = seq(nodes[x].vrec) + mx - 2, *py = seq(nodes[y].vrec) + my - 2;
            for (int32_t i = 0; i < maxRange; i++) { if (*px < *py) return true; if (*px > *py) return false; px--; py--; }
            return mx > my;
        };
        auto popBack = [&]() {
            if (dupAtBack) dupAtBack = false;
            else if (numDummies > 0) numDummies--;
            else win.pop_back();
        };
        if (auxRoot >= 0) {
            popBack();
            const int32_t v = nodes[auxRoot].vrec;
            if (!pre) win.push_front(WinEntry{seq(v), auxRoot, (uint16_t)seqLen(v), (uint16_t)minimPos(v)});
            roots.push_back(auxRoot);
        }
        for (size_t place = 0; place < order.size(); ++place) {
            const int32_t cur = order[place];
            Node& curNode = nodes[cur];
            const int32_t v = curNode.vrec;
            const uint8_t* rs = seq(v); const uint32_t rl = seqLen(v); const int32_t rm = (int32_t)minimPos(v);
            if (!pre) popBack();
            int32_t encodeThreshold = par.encodeThreshold == 0 ? (int32_t)(rl / 2) : par.encodeThreshold;
            // FindBestLzMatch (ReadsClassifier.cpp:55-83)
            MatchResult mr; mr.cost = encodeThreshold + 1;
            bool stop = false;
            int32_t bestNode = -1; bool bestIsReal = false; uint32_t bestLen = 256u;
            if (pre) {
                // the device searched the window (matcher.hip): the cheapest slot, the first among equals
                const fsdev::MatchRow& row = preRows[place];
                mr.cost = row.cost; mr.shift = row.shift; mr.noMismatches = row.no_mismatches != 0;
                if (row.match >= 0) {
                    bestIsReal = true;
                    bestNode = row.match == tableAux ? auxRoot : order.at((size_t)(row.match - tableFirst));
                    bestLen = seqLen(nodes[bestNode].vrec);
                }
            } else {
#if defined(__SSE2__)
            {
                // Eight entries per step are tested for |minPos - rm| * shiftCost <= best cost so far (and <= 127): an entry
                // that fails cannot improve the match (updateLzMatch rejects it on the same test; the best cost only
                // falls), so only the survivors are compared, in window order.
                const uint32_t n = win.size(), cap = win.mask + 1;
                const __m128i vrm = _mm_set1_epi16((int16_t)rm);
                uint32_t i = 0;
                while (i < n && !stop) {
                    const uint32_t pos = (win.head + i) & win.mask;
                    const uint32_t run = std::min(n - i, cap - pos);                 // contiguous stretch of the ring
                    uint32_t k = 0;
                    for (; k + 8 <= run && !stop; k += 8) {
                        int32_t bound = par.shiftCost > 0 ? mr.cost / par.shiftCost : 127; if (bound > 127) bound = 127;      // -s0: shifts are free
                        const __m128i d = _mm_sub_epi16(_mm_loadu_si128((const __m128i*)(win.mp.data() + pos + k)), vrm);
                        const __m128i ad = _mm_max_epi16(d, _mm_sub_epi16(_mm_setzero_si128(), d));
                        uint32_t pass = (uint32_t)_mm_movemask_epi8(_mm_cmpgt_epi16(_mm_set1_epi16((int16_t)(bound + 1)), ad));   // two bits per entry
                        while (pass) {
                            const uint32_t j = (uint32_t)__builtin_ctz(pass) >> 1; pass &= ~(3u << (2 * j));
                            const WinEntry& lz = win.buf[pos + k + j];
                            if (!updateLzMatch(mr, rs, rl, rm, lz.seq, lz.seqLen, lz.minPos)) continue;
                            mr.prevId = (int32_t)(i + k + j);
                            if (mr.cost == 0) { stop = true; break; }
                        }
                    }
                    for (; k < run && !stop; ++k) {
                        const WinEntry& lz = win.buf[pos + k];
                        if (!updateLzMatch(mr, rs, rl, rm, lz.seq, lz.seqLen, lz.minPos)) continue;
                        mr.prevId = (int32_t)(i + k);
                        if (mr.cost == 0) stop = true;
                    }
                    i += run;
                }
            }
#else
            for (uint32_t i = 0; i < win.size(); ++i) {
                const WinEntry& lz = win[i];
                if (!updateLzMatch(mr, rs, rl, rm, lz.seq, lz.seqLen, lz.minPos)) continue;
                mr.prevId = (int32_t)i;
                if (mr.cost == 0) { stop = true; break; }
            }
#endif
            if (!stop && numDummies > 0) {
                // dummy entries (256 x 'N', minPos 0) can only win for reads that are >= 1/3 'N' or with a manual
                // threshold; all dummies are identical, so the first one decides
                if (updateLzMatch(mr, rs, rl, rm, dummy, 256, 0)) mr.prevId = (int32_t)win.size();
            }
            bestIsReal = (uint32_t)mr.prevId < win.size();
            bestLen = bestIsReal ? win[mr.prevId].seqLen : 256u;
            // (a dummy that wins leaves the reference with a null node; the slot the ring holds there is kept as it was)
            bestNode = bestIsReal || win.mask ? win[mr.prevId].node : -1;
            }
            bool identical = (mr.cost == 0 && bestLen == rl);
            bool isHard = mr.cost > encodeThreshold;
            if (identical) identical = bestIsReal && nodes[bestNode].type != TYPE_NONE;
            if (traceRows) {
                fsdev::MatchRow& row = traceRows[place];
                const bool any = mr.cost <= encodeThreshold;
                row.match = !any ? -1 : (!bestIsReal ? -2 : (bestNode == auxRoot ? tableAux : tableFirst + placeOf[bestNode]));
                row.cost = (int16_t)mr.cost; row.shift = (int16_t)(any ? mr.shift : 0); row.no_mismatches = any && mr.noMismatches ? 1 : 0;
                row.identical = identical ? 1 : 0; row.dummy = any && !bestIsReal ? 1 : 0; row.pad = 0;
            }
            if (pre && identical != (preRows[place].identical != 0)) throw std::runtime_error("device matcher: duplicate flags disagree with the tree builder");
            const WinEntry newLz{rs, cur, (uint16_t)rl, (uint16_t)rm};
            if (identical) {
                curNode.type = TYPE_NONE; curNode.lzVrec = -1; curNode.parent = -1;
                Node& parent = nodes[bestNode];
                if (curNode.hasEm) {
                    if (!parent.hasEm) { parent.hasEm = true; parent.em = std::move(curNode.em); }
                    else parent.em.insert(parent.em.end(), curNode.em.begin(), curNode.em.end());
                    curNode.em.clear(); curNode.hasEm = false;
                }
                if (!curNode.trees.empty()) {
                    parent.trees.insert(parent.trees.end(), curNode.trees.begin(), curNode.trees.end());
                    curNode.trees.clear();
                }
                parent.hasEm = true;
                parent.em.push_back(v);
                dupAtBack = true;                 // lzBuffer.push_back(newLz): recycled by the next read
            } else {
                int32_t parentNode = -1;
                PrefixSet* rpb = nullptr; PrefixSet::Pos lbPos{0, 0}; bool haveLb = false;      // lower bound of `cur` in *rpb once known
                if (usePrefix) {
                    const uint32_t expensiveLzThreshold = (uint32_t)encodeThreshold / 2;
                    const bool searchRev = isHard || (par.extraReduceExpensiveLzMatches && mr.cost > (int32_t)expensiveLzThreshold);
                    if (!isHard) encodeThreshold = (int32_t)expensiveLzThreshold;
                    if (rm >= 8) {
                        int32_t bi = dnaToIdxAcgtn(rs[rm - 2]);
                        bi = bi * 5 + dnaToIdxAcgtn(rs[rm - 1]);
                        rpb = &rp[bi];
                        makeKey(cur);
                    }
                    if (searchRev && rm >= 8) {
                        MatchResult fwd, rev; int32_t fwdNode = -1, revNode = -1;
                        fwd.cost = encodeThreshold + 1; rev.cost = encodeThreshold + 1;
                        lbPos = rpb->lowerBound(cur, prefixLess); haveLb = true;
                        const uint32_t maxCnt = W / 2 + 1;
                        rpb->forward(lbPos, maxCnt, prefFwd); rpb->backward(lbPos, maxCnt, prefRev);
                        const NodeSeq* const nsq = nodeSeq.data();
                        {
                            const int32_t* const ids = prefFwd.data(); const size_t nIds = prefFwd.size();
                            for (size_t p = 0; p < nIds; ++p) {
                                if (p + 6 < nIds) _mm_prefetch((const char*)(nsq[ids[p + 6]].seq + rm), _MM_HINT_T0);
                                if (p + 12 < nIds) _mm_prefetch((const char*)&nsq[ids[p + 12]], _MM_HINT_T0);
                                const NodeSeq& c = nsq[ids[p]];
                                if (!updateLzMatch(fwd, rs, rl, rm, c.seq, c.len, (int32_t)c.minPos)) continue;
                                fwdNode = ids[p];
                            }
                        }
                        {
                            const int32_t* const ids = prefRev.data(); const size_t nIds = prefRev.size();
                            for (size_t p = 0; p < nIds; ++p) {
                                if (p + 6 < nIds) _mm_prefetch((const char*)(nsq[ids[p + 6]].seq + rm), _MM_HINT_T0);
                                if (p + 12 < nIds) _mm_prefetch((const char*)&nsq[ids[p + 12]], _MM_HINT_T0);
                                const NodeSeq& c = nsq[ids[p]];
                                if (!updateLzMatch(rev, rs, rl, rm, c.seq, c.len, (int32_t)c.minPos)) continue;
                                revNode = ids[p];
                            }
                        }
                        const int32_t minCost = fwd.cost < rev.cost ? fwd.cost : rev.cost;
                        if (minCost < encodeThreshold && minCost < mr.cost) {
                            if (fwd.cost < rev.cost) { parentNode = fwdNode; mr = fwd; } else { parentNode = revNode; mr = rev; }
                            isHard = false;
                        }
                    }
                }
                if (isHard) {
                    curNode.type = TYPE_HARD; curNode.lzVrec = -1; curNode.parent = -1;
                    roots.push_back(cur);
                } else {
                    if (parentNode < 0) parentNode = bestNode;
                    curNode.type = TYPE_LZ; curNode.parent = parentNode; curNode.lzVrec = nodes[parentNode].vrec;
                    curNode.shift = (int16_t)mr.shift; curNode.noMismatches = mr.noMismatches; curNode.cost = (int16_t)mr.cost;
                    nodes[parentNode].children.push_back(cur);
                }
                if (!pre) win.push_front(newLz);
                if (rpb) {                        // std::set::insert: skipped when an equivalent node is present
                    const PrefixSet::Pos it = haveLb ? lbPos : rpb->lowerBound(cur, prefixLess);
                    if (rpb->atEnd(it) || prefixLess(cur, rpb->at(it))) rpb->insert(it, cur);
                }
            }
        }
    }
    static int32_t dnaToIdxAcgtn(uint8_t c) { switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case 'N': return 4; } return -1; }

    // ------------------------------------------------------------------------------------------
    // ContigBuilder (fastore_pack/ContigBuilder.cpp:50-669, costs ContigBuilder.h:128-151)
    struct WorkNode { int32_t match = -1; std::vector<uint16_t> newVariantPositions; };
    struct BuildInfo {
        Contig cons; int32_t mainNode = -1;
        std::vector<WorkNode> nodes; std::vector<uint16_t> variantFreqPerPos, recordsPerPos;
        std::vector<int32_t> removedNodes;
        void reset(uint32_t L)
        {
            cons.sequence.assign(2 * L, '.'); cons.variant.assign(2 * L, 0); cons.readLen = L;
            variantFreqPerPos.assign(2 * L, 0); recordsPerPos.assign(2 * L, 0);
            mainNode = -1; nodes.clear(); removedNodes.clear();
            cons.variantsCount = 0; cons.rangeFirst = L; cons.rangeSecond = L;
        }
    };
    void addChildrenToQueue(std::deque<int32_t>& q, int32_t n)
    {
        const auto& ch = nodes[n].children;
        if (ch.size() > 1) {
            std::vector<int32_t> c2;
            for (int32_t c : ch) { if (nodes[c].children.empty()) q.push_back(c); else c2.push_back(c); }
            q.insert(q.end(), c2.begin(), c2.end());
        } else q.push_back(ch.front());
    }
    float normalEncodeCost(const Node& n) const
    {
        float rleCost = 0.0f;
        const int32_t as = n.shift < 0 ? -n.shift : n.shift;
        if (as != n.cost) rleCost = 1.0f + n.cost / 1.5f;
        return (1.0f + n.cost) + rleCost + 2.0f;
    }
    bool consCostExceeds(const BuildInfo& bi, const WorkNode& w, uint32_t ham, const Node& n) const
    {
        uint32_t newVarCost = 0;
        for (uint16_t p : w.newVariantPositions) newVarCost += bi.recordsPerPos[p];
        if (newVarCost > 0) ham -= 1;
        // "(float)(...) + float(newVarCost) * 0.9" is evaluated in double and rounded to the float return type
        const float c = (float)((double)(float)(1 + ham + par.beginCut + par.endCut) + (double)(float)newVarCost * 0.9);
        return c > normalEncodeCost(n);
    }
    void updateRange(BuildInfo& bi, uint32_t m, uint32_t consBegin, uint32_t consEnd, uint32_t L, bool first)
    {
        const uint32_t f = (m <= par.beginCut) ? consBegin + m + sigLen : consBegin + par.beginCut;
        const uint32_t s = (L - m - sigLen <= par.endCut) ? consBegin + m : consEnd - par.endCut;
        if (first) { bi.cons.rangeFirst = f; bi.cons.rangeSecond = s; }
        else { bi.cons.rangeFirst = std::min(bi.cons.rangeFirst, f); bi.cons.rangeSecond = std::max(bi.cons.rangeSecond, s); }
    }
    // consensus update of one member read: unset positions take its base, set positions that differ count a variant,
    // every covered position counts the record (ContigBuilder.cpp:232-244)
    void mergeIntoConsensus(BuildInfo& bi, const uint8_t* s, uint32_t consBegin, uint32_t L)
    {
        uint32_t i = par.beginCut; const uint32_t end = L - par.endCut;
        char* cs = bi.cons.sequence.data() + consBegin;
        uint16_t* rpp = bi.recordsPerPos.data() + consBegin; uint16_t* vf = bi.variantFreqPerPos.data() + consBegin;
#if defined(__SSE2__)
        const __m128i dot = _mm_set1_epi8('.'), one16 = _mm_set1_epi16(1);
        for (; i + 16 <= end; i += 16) {
            const __m128i c = _mm_loadu_si128((const __m128i*)(cs + i)), r = _mm_loadu_si128((const __m128i*)(s + i));
            const __m128i isDot = _mm_cmpeq_epi8(c, dot);
            _mm_storeu_si128((__m128i*)(cs + i), _mm_or_si128(_mm_and_si128(isDot, r), _mm_andnot_si128(isDot, c)));
            uint32_t diff = ((uint32_t)_mm_movemask_epi8(_mm_or_si128(isDot, _mm_cmpeq_epi8(c, r))) ^ 0xFFFFu) & 0xFFFFu;   // set and different
            while (diff) { const uint32_t j = (uint32_t)__builtin_ctz(diff); diff &= diff - 1; vf[i + j]++; }
            _mm_storeu_si128((__m128i*)(rpp + i), _mm_add_epi16(_mm_loadu_si128((const __m128i*)(rpp + i)), one16));
            _mm_storeu_si128((__m128i*)(rpp + i + 8), _mm_add_epi16(_mm_loadu_si128((const __m128i*)(rpp + i + 8)), one16));
        }
#endif
        for (; i < end; ++i) {
            if (cs[i] == '.') cs[i] = (char)s[i];
            else if (cs[i] != (char)s[i]) vf[i]++;
            rpp[i]++;
        }
    }
    bool addRecord(BuildInfo& bi, int32_t n, bool fullMatchOnly)
    {
        const Node& node = nodes[n];
        if (!node.trees.empty()) return false;                  // AvoidTreesInConsensus
        const uint32_t L = bi.cons.readLen;
        const int32_t v = node.vrec;
        // The consensus frame is laid out for reads of the first member's length; the reference indexes it the same way
        // for every member (ContigBuilder::AddRecord) and reads outside its buffers when lengths differ ("TODO: verify
        // for variable-length reads", FastqCompressor.cpp:1766).  That has no defined result to reproduce: refuse.
        if (seqLen(v) != L) throw std::runtime_error("reads of different lengths in one read cluster: variable-length libraries are not supported (neither by the reference encoder)");
        const uint8_t* s = seq(v); const uint32_t m = minimPos(v);
        const uint32_t consBegin = L - m, consEnd = consBegin + L;
        if (!bi.nodes.empty()) {
            WorkNode w; uint32_t ham = 0;
            {
                uint32_t i = par.beginCut; const uint32_t end = L - par.endCut;
                const char* cs = bi.cons.sequence.data() + consBegin;
#if defined(__SSE2__)
                // 16 positions per step: mismatches against a set consensus base and 'N' on an unset one are both rare, so
                // the per-position work only runs for the flagged lanes (in position order, like the scalar loop)
                const __m128i dot = _mm_set1_epi8('.'), en = _mm_set1_epi8('N');
                for (; i + 16 <= end; i += 16) {
                    const __m128i c = _mm_loadu_si128((const __m128i*)(cs + i)), r = _mm_loadu_si128((const __m128i*)(s + i));
                    const uint32_t isDot = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(c, dot));
                    const uint32_t same = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(c, r));
                    const uint32_t isN = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(r, en));
                    uint32_t flagged = ((~isDot & ~same) | (isDot & isN)) & 0xFFFFu;
                    while (flagged) {
                        const uint32_t j = (uint32_t)__builtin_ctz(flagged); flagged &= flagged - 1;
                        const uint32_t p = consBegin + i + j;
                        if ((isDot >> j) & 1u) return false;           // unset consensus position under an 'N'
                        ham++; if (bi.variantFreqPerPos[p] == 0) w.newVariantPositions.push_back((uint16_t)p);
                    }
                }
#endif
                for (; i < end; ++i) {
                    const uint32_t p = consBegin + i;
                    if (cs[i] != '.' && cs[i] != (char)s[i]) { ham++; if (bi.variantFreqPerPos[p] == 0) w.newVariantPositions.push_back((uint16_t)p); }
                    else if (cs[i] == '.' && s[i] == 'N') return false;
                }
            }
            if (fullMatchOnly) { if (!w.newVariantPositions.empty()) return false; }
            else {
                const uint32_t maxShift = par.maxRecordShiftDifference == 0 ? L / 2 : par.maxRecordShiftDifference;
                const int32_t lastM = (int32_t)minimPos(nodes[bi.nodes.back().match].vrec);
                // the reference's macro is  ABS(x) ((x) >= 0 ? (x) : (-x))  -- applied to "last - cur" the negative
                // branch expands to (-last - cur), i.e. never exceeds maxShift (fastore_bin/Globals.h:70, ContigBuilder.cpp:219)
                int32_t d = lastM - (int32_t)m; if (d < 0) d = -lastM - (int32_t)m;
#ifdef FS_DEBUG_DUMP
                  printf("D ham %u newvars %zu nvc %u d %d maxShift %u normal %f shift %d cost %d exceeds %d\n", ham, w.newVariantPositions.size(), nvc, d, maxShift, normalEncodeCost(node), node.shift, node.cost, (int)consCostExceeds(bi, w, ham, node)); }
#endif
                if (!(w.newVariantPositions.empty() || (ham <= par.maxHammingDistance && w.newVariantPositions.size() <= par.maxNewVariantsPerRead))
                    || d > (int32_t)maxShift || consCostExceeds(bi, w, ham, node))
                    return false;
            }
            bi.cons.variantsCount += (uint32_t)w.newVariantPositions.size();
            mergeIntoConsensus(bi, s, consBegin, L);
            updateRange(bi, m, consBegin, consEnd, L, false);
            w.match = n; bi.nodes.push_back(std::move(w));
        } else {
            if (memchr(s, 'N', L) != nullptr) return false;
            for (uint32_t i = par.beginCut; i < L - par.endCut; ++i) bi.cons.sequence[consBegin + i] = (char)s[i];
            updateRange(bi, m, consBegin, consEnd, L, true);
            WorkNode w; w.match = n; bi.nodes.push_back(std::move(w));
        }
        return true;
    }
    void optimizeContig(BuildInfo& bi)
    {
        std::vector<uint32_t> idxToRemove;
        for (uint32_t i = 0; i < bi.nodes.size(); ++i) {
            const WorkNode& n = bi.nodes[i];
            if (n.newVariantPositions.empty()) continue;
            bool onlyOne = false;
            for (uint16_t pos : n.newVariantPositions) if (bi.variantFreqPerPos[pos] == 1) onlyOne = true;
            if (onlyOne) idxToRemove.push_back(i);
        }
        if (idxToRemove.empty()) return;
        std::vector<WorkNode> old = std::move(bi.nodes);
        const uint32_t L = bi.cons.readLen;
        bi.reset(L);
        uint32_t idx = 0;
        for (uint32_t r : idxToRemove) {
            if (r > idx) for (uint32_t k = idx; k < r; ++k) bi.nodes.push_back(old[k]);
            bi.removedNodes.push_back(old[r].match);
            idx = r + 1;
        }
        for (uint32_t k = idx; k < old.size(); ++k) bi.nodes.push_back(old[k]);
        {
            const int32_t v = nodes[bi.nodes[0].match].vrec; const uint8_t* s = seq(v); const uint32_t m = minimPos(v);
            const uint32_t consBegin = L - m, consEnd = consBegin + L;
            for (uint32_t i = par.beginCut; i < L - par.endCut; ++i) bi.cons.sequence[consBegin + i] = (char)s[i];
            updateRange(bi, m, consBegin, consEnd, L, true);
        }
        for (size_t k = 1; k < bi.nodes.size(); ++k) {
            WorkNode& w = bi.nodes[k];
            const int32_t v = nodes[w.match].vrec; const uint8_t* s = seq(v); const uint32_t m = minimPos(v);
            const uint32_t consBegin = L - m, consEnd = consBegin + L;
            w.newVariantPositions.clear();
            for (uint32_t i = par.beginCut; i < L - par.endCut; ++i) {
                const uint32_t p = consBegin + i;
                if (bi.cons.sequence[p] != '.' && bi.cons.sequence[p] != (char)s[i]) if (bi.variantFreqPerPos[p] == 0) w.newVariantPositions.push_back((uint16_t)p);
            }
            bi.cons.variantsCount += (uint32_t)w.newVariantPositions.size();
            mergeIntoConsensus(bi, s, consBegin, L);
            updateRange(bi, m, consBegin, consEnd, L, false);
        }
    }
    static void removeChild(std::vector<int32_t>& ch, int32_t c) { auto it = std::find(ch.begin(), ch.end(), c); if (it != ch.end()) ch.erase(it); }
    void updateContigLinkage(BuildInfo& bi)
    {
        std::vector<int32_t> consNodes;
        for (const auto& w : bi.nodes) consNodes.push_back(w.match);
        std::sort(consNodes.begin(), consNodes.end());
        auto inCons = [&](int32_t n) { return std::binary_search(consNodes.begin(), consNodes.end(), n); };
        int32_t bestParent = -1; int32_t minCost = 255;
        for (auto& w : bi.nodes) {
            Node& n = nodes[w.match];
            if (!inCons(n.parent)) {
                const int32_t parent = n.parent;
                removeChild(nodes[parent].children, w.match);
                n.parent = -1;
                if (std::find(bi.removedNodes.begin(), bi.removedNodes.end(), parent) == bi.removedNodes.end()
                    && (bi.mainNode < 0 || minCost > n.cost)) {
                    bestParent = parent; bi.mainNode = w.match; minCost = n.cost;
                }
            }
        }
        if (bi.mainNode < 0 || bestParent < 0) throw std::runtime_error("contig linkage: no external parent");
        nodes[bi.mainNode].parent = bestParent;
        nodes[bestParent].children.push_back(bi.mainNode);
        for (size_t k = 0; k < bi.nodes.size(); ++k) if (bi.nodes[k].match == bi.mainNode) { bi.nodes.erase(bi.nodes.begin() + k); break; }
        consNodes.erase(std::lower_bound(consNodes.begin(), consNodes.end(), bi.mainNode));
        {
            auto& ch = nodes[bi.mainNode].children;
            for (size_t k = 0; k < ch.size();) { if (inCons(ch[k])) ch.erase(ch.begin() + k); else ++k; }
        }
        for (auto& w : bi.nodes) {
            Node& n = nodes[w.match];
            if (n.children.empty()) continue;                  // children == NULL (an emptied list is deleted by RemoveChild)
            for (int32_t c : n.children) if (!inCons(c)) { nodes[c].parent = bi.mainNode; nodes[bi.mainNode].children.push_back(c); }
            n.children.clear();
        }
    }
    void postProcessContig(BuildInfo& bi)
    {
        const uint32_t L = bi.cons.readLen;
        if (bi.cons.variantsCount != 0) {
            auto agctn = [](uint8_t c) -> int { switch (c) { case 'A': return 0; case 'G': return 1; case 'C': return 2; case 'T': return 3; case 'N': return 4; } return 0; };
            static const char idxToDna[] = "AGCTN";
            std::map<uint32_t, std::vector<uint32_t>> posStats;
            for (const auto& w : bi.nodes) {
                const int32_t v = nodes[w.match].vrec; const uint8_t* s = seq(v);
                const uint32_t consBegin = L - minimPos(v);
                for (uint32_t i = par.beginCut; i < L - par.endCut; ++i) {
                    const uint32_t p = consBegin + i;
                    if (bi.variantFreqPerPos[p] > 0) { auto& st = posStats[p]; if (st.empty()) st.resize(5); st[agctn(s[i])] += 1; }
                }
            }
            for (auto& st : posStats) {
                char c = 'N'; uint32_t maxFreq = 0;
                for (uint32_t i = 0; i < 5; ++i) if (st.second[i] > maxFreq) { maxFreq = st.second[i]; c = idxToDna[i]; }
                bi.cons.sequence[st.first] = c;
            }
            for (uint32_t i = 0; i < bi.variantFreqPerPos.size(); ++i) {
                bi.cons.variant[i] = bi.variantFreqPerPos[i] != 0;
                bi.cons.variantsCount += bi.variantFreqPerPos[i] != 0;
            }
        }
        for (char& c : bi.cons.sequence) if (c == '.') c = 'N';
        // std::sort(nodes) by record minimPos -- unstable, libstdc++ order (ContigBuilder.cpp:521)
        introsort(bi.nodes.data(), bi.nodes.size(), [&](const WorkNode& a, const WorkNode& b) {
            return minimPos(nodes[a.match].vrec) < minimPos(nodes[b.match].vrec);
        });
    }
    void storeContig(BuildInfo& bi)
    {
        const int32_t id = (int32_t)contigs.size();
        contigs.push_back(Contig());
        Contig& c = contigs.back();
        c = std::move(bi.cons);
        nodes[bi.mainNode].contig = id;
        nodes[bi.mainNode].type = TYPE_LZ;
        for (auto& w : bi.nodes) { nodes[w.match].type = TYPE_CONTIG_READ; c.nodes.push_back(w.match); }
    }
    void buildContigs(int32_t root)
    {
        const uint32_t L = seqLen(nodes[root].vrec);
        std::deque<int32_t> nextQueue;
        if (!nodes[root].children.empty()) addChildrenToQueue(nextQueue, root);
        BuildInfo bi;
        while (!nextQueue.empty()) {
            int32_t node = nextQueue.front(); nextQueue.pop_front();
            bi.reset(L);
#ifdef FS_DEBUG_DUMP
#endif
            if (!addRecord(bi, node, false)) { if (!nodes[node].children.empty()) addChildrenToQueue(nextQueue, node); continue; }
            std::deque<int32_t> curQueue = std::move(nextQueue); nextQueue.clear();
            if (!nodes[node].children.empty()) addChildrenToQueue(curQueue, node);
            while (!curQueue.empty()) {
                node = curQueue.front(); curQueue.pop_front();
                { bool r1 = addRecord(bi, node, true);
#ifdef FS_DEBUG_DUMP
#endif
                if (!r1) { nextQueue.push_back(node); continue; } }
                if (false) {}
                else if (!nodes[node].children.empty()) addChildrenToQueue(curQueue, node);
            }
            std::swap(curQueue, nextQueue);
            while (!curQueue.empty()) {
                node = curQueue.front(); curQueue.pop_front();
                { bool r2 = addRecord(bi, node, false);
#ifdef FS_DEBUG_DUMP
#endif
                if (!r2) { nextQueue.push_back(node); continue; } }
                if (false) {}
                else if (!nodes[node].children.empty()) addChildrenToQueue(curQueue, node);
            }
#ifdef FS_DEBUG_DUMP
#endif
            if (bi.nodes.size() < par.minConsensusSize) continue;
            optimizeContig(bi);
#ifdef FS_DEBUG_DUMP
#endif
            if (bi.nodes.size() < par.minConsensusSize) continue;
            updateContigLinkage(bi);
            postProcessContig(bi);
            storeContig(bi);
        }
    }

    // ------------------------------------------------------------------------------------------
    // record-level emitters (fastore_pack/FastqCompressor.cpp:1388-2119)
    void compressId(int32_t v)
    {
        if (!hasHeaders) return;
        const Rec& r = R(v);
        if (!B->headBit.empty()) out->idRefs.push_back(IdRef{B->headBit[vrecs[v].rec], r.headLen});      // tokenised on the device
        else compressReadId(*headp, B->head.data() + r.headOff, r.headLen, out->s[S_IdToken], out->s[S_IdValue]);
        out->rawIdSize += r.headLen;
    }
    bool packedQuality() const { return !B->quaBit.empty(); }
    uint32_t quaBits() const { return cfg.quaParams.method == MET_BINARY ? 1u : (cfg.quaParams.method == MET_8BIN ? 3u : 6u); }
    void refQuality(const uint8_t* s, uint32_t bit, uint32_t len, bool reverse)
    {
        QuaRef r{bit, (uint16_t)len, (uint8_t)(reverse ? 1 : 0), 0, (uint32_t)out->quaN.size()};
        if (cfg.quaParams.method == MET_8BIN || cfg.quaParams.method == MET_BINARY) {      // the scores under an 'N' are left out (QVZ codes them all)
            uint32_t cnt = 0;
            for (const uint8_t* q = (const uint8_t*)memchr(s, 'N', len); q; q = (const uint8_t*)memchr(q + 1, 'N', (size_t)(s + len - q - 1))) { out->quaN.push_back((uint8_t)(q - s)); ++cnt; }
            r.nCount = (uint8_t)cnt;                                 // len <= 255, and a read of nothing but 'N' has no signature
        }
        out->quaRefs.push_back(r); out->quaSymbols += len - r.nCount;
    }
    void compressQuality(int32_t v)
    {
        if (packedQuality()) { refQuality(seq(v), B->quaBit[vrecs[v].rec], seqLen(v), isReverse(v)); return; }
        compressReadQuality(cfg, seq(v), qua(v), seqLen(v), isReverse(v), out->s[S_Quality], qvzp, &well);
    }

    void compressHardRead(int32_t v)
    {
        putSym(S_Rev, isReverse(v));
        putByte(S_Flag, ReadDifficult);
        const uint8_t* s = seq(v); const int32_t L = (int32_t)seqLen(v), m = (int32_t)minimPos(v);
        if (devEmit) { pushOp(fsdev::EMIT_HARD, seqRel(s), 0, (uint32_t)L, 0, (uint32_t)m, 0, 0, 0); out->emitBound[fsdev::ECH_HARD] += (uint32_t)L + 1u; return; }
        for (int32_t i = 0; i < L; ++i) {
            if (i < m || i >= m + (int32_t)sigLen) putByte(S_HardReads, s[i]);
            else if (i == m) putByte(S_HardReads, '.');
        }
    }
    void compressExactRead(int32_t v) { putSym(S_Rev, isReverse(v)); putByte(S_Flag, ReadIdentical); }

    void compressNormalMatch(const Node& n, uint32_t lzId, bool expensive)
    {
        const int32_t v = n.vrec, lv = n.lzVrec;
        putSym(S_Rev, isReverse(v));
        putByte(S_Shift, (uint32_t)(ShiftOffset + n.shift));
        if (devEmit) out->lzIds.push_back(lzId); else lzRle0.put(lzId);
        const int32_t flag = n.noMismatches ? ReadShiftOnly : (expensive ? ReadFullExpensive : ReadFullEncode);
        putByte(S_Flag, (uint32_t)flag);
        const uint8_t* bestSeq = seq(lv); uint32_t bestLen = seqLen(lv), bestPos = minimPos(lv);
        const uint8_t* newSeq = seq(v); uint32_t newLen = seqLen(v);
        if (devEmit) {
            const uint32_t mode = flag == ReadFullEncode ? fsdev::EMIT_FULL : (flag == ReadFullExpensive ? fsdev::EMIT_EXPENSIVE : fsdev::EMIT_SHIFT_ONLY);
            pushOp(fsdev::EMIT_MATCH, seqRel(newSeq), seqRel(bestSeq), newLen, bestLen, 0, bestPos, n.shift, mode);
            out->emitBound[fsdev::ECH_LETTERS] += newLen;
            if (mode == fsdev::EMIT_FULL) out->emitBound[fsdev::ECH_MATCH_BITS] += newLen; else if (mode == fsdev::EMIT_EXPENSIVE) out->emitBound[fsdev::ECH_MATCH_BIN] += newLen;
            out->emitBound[fsdev::ECH_COUNT] += 6u;
            return;
        }
        if (n.shift >= 0) { bestSeq += n.shift; bestLen -= n.shift; bestPos -= n.shift; }
        else {
            for (int32_t i = 0; i < -n.shift; ++i) putSym(S_LettersX, d2i(newSeq[i]), d2i('N'));
            newSeq += -n.shift; newLen -= -n.shift;
        }
        const uint32_t minLen = bestLen < newLen ? bestLen : newLen;
        if (flag == ReadFullEncode) {
            for (uint32_t i = 0; i < minLen; ++i) {
                if (i == bestPos) { i += sigLen - 1; continue; }
                if (bestSeq[i] == newSeq[i]) matchRle.put(true);
                else { matchRle.put(false); putSym(S_LettersX, d2i(newSeq[i]), d2i(bestSeq[i])); }
            }
        } else if (flag == ReadFullExpensive) {
            for (uint32_t i = 0; i < minLen; ++i) {
                if (i == bestPos) { i += sigLen - 1; continue; }
                putSym(S_MatchBinary, bestSeq[i] == newSeq[i]);
                if (bestSeq[i] != newSeq[i]) putSym(S_LettersX, d2i(newSeq[i]), d2i(bestSeq[i]));
            }
        }
        for (uint32_t i = minLen; i < newLen; ++i) putSym(S_LettersX, d2i(newSeq[i]), d2i('N'));
    }
    void compressContigRead(int32_t v, bool useTreeShift)
    {
        ConsEnc& ce = consStack.back();
        const int32_t m = (int32_t)minimPos(v);
        const int32_t dpos = m - ce.lastMinimPos; ce.lastMinimPos = m;
        const uint32_t stream = useTreeShift ? S_TreeShift : S_CShift;
        const uint32_t readLen = seqLen(v);
        if (readLen * 2 >= 256) putByte(stream, (uint32_t)m); else putByte(stream, (uint32_t)(ShiftOffset + dpos));
        putSym(S_Rev, isReverse(v));
        const uint32_t consStart = readLen - m;
        const uint8_t* s = seq(v); const Contig& def = *ce.def;
        if (readLen != def.readLen) throw std::runtime_error("reads of different lengths in one read cluster: variable-length libraries are not supported (neither by the reference encoder)");
        if (devEmit) { pushOp(fsdev::EMIT_CREAD, seqRel(s), ce.defOff, readLen, 0, (uint32_t)m, def.readLen, 0, 0); out->emitBound[fsdev::ECH_CLETTERS] += readLen; return; }
        uint32_t it = 0;
        while (it < par.beginCut) {
            if (it == (uint32_t)m) { it += sigLen; continue; }
            putSym(S_CLetters, d2i(s[it]), d2i((uint8_t)def.sequence[consStart + it])); it++;
        }
        while (it < readLen - par.endCut) {
            if (it == (uint32_t)m) { it += sigLen; continue; }
            if (def.variant[consStart + it]) putSym(S_CLetters, d2i(s[it]), d2i((uint8_t)def.sequence[consStart + it]));
            it++;
        }
        while (it < readLen) { putSym(S_CLetters, d2i(s[it]), d2i((uint8_t)def.sequence[consStart + it])); it++; }
    }
    // LzCompressorSE::CompressRead / LzCompressorPE::CompressRead
    void compressRead(int32_t v, ReadMatchType type, bool aux = false)
    {
        compressId(v);
        switch (type) {
        case HardRead: compressHardRead(v); break;
        case ExactMatch: compressExactRead(v); break;
        case ContigRead: compressContigRead(v, aux); break;
        default: break;
        }
        compressQuality(v);
        if (pe) compressPair(v, type);
    }
    // LzCompressorSE::CompressMatch / LzCompressorPE::CompressMatch
    void compressMatch(int32_t n)
    {
        const Node& node = nodes[n];
        auto& hist = lzStack.back().history;
        uint32_t lzId = 0;
        for (size_t k = hist.size(); k > 0; --k, ++lzId) if (hist[k - 1] == node.lzVrec) break;
        if (lzId == hist.size()) throw std::runtime_error("LZ parent not found in history");
        int32_t as = node.shift * par.shiftCost; if (as < 0) as = -as;
        const uint32_t mism = (uint32_t)(node.cost - as) / (uint32_t)par.mismatchCost;
        compressNormalMatch(node, lzId, mism > par.maxMismatchesLowCost);
        hist.push_back(node.vrec);
        compressId(node.vrec);
        compressQuality(node.vrec);
        if (pe) compressPair(node.vrec, LzMatch);
    }
    void compressExactChildren(int32_t n) { for (size_t k = 0; k < nodes[n].em.size(); ++k) compressRead(nodes[n].em[k], ExactMatch); }

    void storeContigDefinition(const Contig& d, const uint8_t* mainSeq, uint32_t mainSigPos)
    {
        (void)mainSeq;
        const uint32_t lzFirst = d.readLen - mainSigPos, lzSecond = lzFirst + d.readLen;
        if (d.readLen < 128) {
            const int32_t r1 = ShiftOffset - (int32_t)d.readLen / 2, r2 = ShiftOffset - (int32_t)d.readLen * 3 / 2;
            putByte(S_TreeShift, (uint32_t)((int32_t)d.rangeFirst + r1));
            putByte(S_TreeShift, (uint32_t)((int32_t)d.rangeSecond + r2));
        } else {
            putByte(S_TreeShift, d.rangeFirst);
            const int32_t rangeDiff = (int32_t)d.rangeSecond - (int32_t)d.rangeFirst;
            const int32_t rescale = (int32_t)sigLen + 2 + 2;      // Default::BeginCut + Default::EndCut
            putByte(S_TreeShift, (uint32_t)(rangeDiff - (int32_t)d.readLen + rescale));
        }
        if (devEmit) {
            ConsEnc& ce = consStack.back();
            ce.defOff = (uint32_t)out->contigBytes.size();
            out->contigBytes.insert(out->contigBytes.end(), (const uint8_t*)d.sequence.data(), (const uint8_t*)d.sequence.data() + d.sequence.size());
            out->contigBytes.insert(out->contigBytes.end(), d.variant.begin(), d.variant.end());
            if (d.sequence.size() != 2u * d.readLen || d.variant.size() != 2u * d.readLen) throw std::runtime_error("contig of an unexpected extent");
            pushOp(fsdev::EMIT_CDEF, 0, ce.defOff, d.rangeFirst, d.rangeSecond, mainSigPos, d.readLen, 0, 0);
            out->emitBound[fsdev::ECH_CLETTERS] += 2u * d.readLen; out->emitBound[fsdev::ECH_CMATCH_BITS] += 2u * d.readLen;
            return;
        }
        for (uint32_t i = d.rangeFirst; i < d.rangeSecond; ++i) {
            if (i == d.readLen) { i += sigLen - 1; continue; }
            consMatchRle.put(d.variant[i] == 0);
            if (i < lzFirst + 2 || i >= lzSecond - 2 || d.variant[i] != 0) putSym(S_CLetters, d2i((uint8_t)d.sequence[i]), d2i('N'));
        }
    }
    void compressContig(int32_t n)
    {
        const Contig& c = contigs[nodes[n].contig];
        consStack.push_back(ConsEnc());
        consStack.back().def = &c; consStack.back().lastMinimPos = (int32_t)minimPos(nodes[n].vrec);
        putByte(S_Flag, ReadContigGroupStart);
        storeContigDefinition(c, seq(nodes[n].vrec), minimPos(nodes[n].vrec));
        bool first = true;
        for (int32_t mn : c.nodes) {
            if (!first) putByte(S_Flag, ReadContigGroupNext);
            compressRead(nodes[mn].vrec, ContigRead, first);
            first = false;
            lzStack.back().history.push_back(nodes[mn].vrec);
            if (nodes[mn].hasEm) compressExactChildren(mn);
            if (!nodes[mn].trees.empty()) compressSubTree(mn);
        }
        putByte(S_Flag, ReadGroupEnd);
        consStack.pop_back();
    }
    void compressNode(int32_t n)
    {
        if (nodes[n].type == TYPE_HARD) {
            lzStack.back().history.clear();
            compressRead(nodes[n].vrec, HardRead);
            lzStack.back().history.push_back(nodes[n].vrec);
            if (nodes[n].hasEm) compressExactChildren(n);
            if (!nodes[n].trees.empty()) compressSubTree(n);
        } else {
            compressMatch(n);
            if (nodes[n].hasEm) compressExactChildren(n);
            if (!nodes[n].trees.empty()) compressSubTree(n);
            if (nodes[n].contig >= 0) compressContig(n);
        }
    }
    void encodeTree(int32_t root, bool skipRoot)
    {
        std::deque<int32_t> nq;
        if (skipRoot) nq.insert(nq.end(), nodes[root].children.begin(), nodes[root].children.end());
        else nq.push_back(root);
        while (!nq.empty()) {
            const int32_t n = nq.front(); nq.pop_front();
            compressNode(n);
            nq.insert(nq.end(), nodes[n].children.begin(), nodes[n].children.end());
        }
    }
    void compressSubTree(int32_t n)
    {
        const int32_t rootV = nodes[n].vrec;
        const std::vector<uint32_t> treeList = nodes[n].trees;      // copy: nodes may reallocate below
        for (uint32_t t : treeList) {
            const TreeIn& tree = G->trees[t];
            consStack.push_back(ConsEnc());
            consStack.back().lastMinimPos = tree.mainSignaturePos;
            const uint32_t rl = seqLen(rootV);
            if (rl * 2 >= 256) putByte(S_TreeShift, (uint32_t)tree.mainSignaturePos);
            else putByte(S_TreeShift, (uint32_t)(ShiftOffset + (tree.mainSignaturePos - (int32_t)minimPos(rootV))));
            putByte(S_Flag, ReadTreeGroupStart);
            lzStack.push_back(LzContext());
            // local copy of the root with the sub-tree's signature position
            const int32_t localV = (int32_t)vrecs.size();
            vrecs.push_back(VRec{vrecs[rootV].rec, (uint16_t)tree.mainSignaturePos});
            const int32_t localRoot = (int32_t)nodes.size();
            nodes.push_back(Node());
            nodes[localRoot].vrec = localV; nodes[localRoot].type = TYPE_NONE;
            lzStack.back().history.push_back(localV);
            std::vector<int32_t> order(tree.nodeCount);
            for (uint32_t k = 0; k < tree.nodeCount; ++k) order[k] = (int32_t)(tree.nodeBegin - nodeBase + k);
            std::vector<int32_t> roots;
            constructMatchTree(order, roots, localRoot, t < callOfTree.size() ? callOfTree[t] : -1);
            const size_t contigMark = contigs.size();
            bool isFirst = true;
            for (int32_t root : roots) {
                if (!nodes[root].children.empty()) buildContigs(root);
                encodeTree(root, isFirst);
                isFirst = false;
            }
            (void)contigMark;
            putByte(S_Flag, ReadGroupEnd);
            consStack.pop_back();
            lzStack.pop_back();
        }
    }

    // ------------------------------------------------------------------------------------------
    // PE mate coder -- filled in by frontend_pe.inc
    void compressPair(int32_t v, ReadMatchType seType);
    void resetPair();
    // the pairs of the bin in the order the walk met them; their searches and mate streams come behind the walk (frontend_pe.inc)
    struct PairTodo { int32_t v; uint8_t seType; };
    std::vector<PairTodo> pairTodo;
    MateFn mateMatcher;                                   // the device's mate search (empty: the host's)
    std::vector<fsdev::MateRow>* mateTrace = nullptr;     // parity harness: the host search's rows
    std::vector<fsdev::MatePair> matePairs; std::vector<fsdev::MateRow> mateRows;
    int32_t pairThreshold(const Rec& r) const { return par.pairEncodeThreshold == 0 ? (int32_t)(r.seqLen / 1.5) : par.pairEncodeThreshold; }
    fsdev::MateRow searchPairHost(int32_t v, int32_t idx);
    void emitPair(int32_t v, ReadMatchType seType, const fsdev::MateRow& row, int32_t matchV);
    void finishPairs();
    struct PairState;
    PairState* pairState = nullptr;

    // ------------------------------------------------------------------------------------------
    void initNodes(const BinIn& bin)
    {
        // node table: Batch nodes [nodeBase, nodeEnd) of this bin; bins own contiguous node ranges
        if (bin.topCount == 0 || bin.recCount == 0 || (size_t)bin.topBegin + bin.topCount > G->topNodes.size()) throw std::runtime_error("Corrupted bin: no records in a standard bin");
        nodeBase = G->topNodes[bin.topBegin];
        uint32_t nodeEnd = nodeBase;
        for (uint32_t k = 0; k < bin.topCount; ++k) nodeEnd = std::max(nodeEnd, G->topNodes[bin.topBegin + k] + 1);
        // sub-tree nodes are allocated after their owner, so scan trees reachable from the range
        for (uint32_t i = nodeBase; i < nodeEnd; ++i) {
            const NodeIn& ni = G->nodes[i];
            for (uint32_t t = 0; t < ni.treeCount; ++t) nodeEnd = std::max(nodeEnd, G->trees[ni.treeBegin + t].nodeBegin + G->trees[ni.treeBegin + t].nodeCount);
        }
        recBase = bin.recBegin;
        vrecs.resize(bin.recCount);
        for (uint32_t i = 0; i < bin.recCount; ++i) vrecs[i] = VRec{recBase + i, B->recs[recBase + i].minimPos};
        nodes.clear(); nodes.resize(nodeEnd - nodeBase);
        for (uint32_t i = nodeBase; i < nodeEnd; ++i) {
            const NodeIn& ni = G->nodes[i]; Node& n = nodes[i - nodeBase];
            n.vrec = (int32_t)(ni.rec - recBase);
            if (ni.emCount) { n.hasEm = true; n.em.resize(ni.emCount); for (uint32_t k = 0; k < ni.emCount; ++k) n.em[k] = (int32_t)(G->emRecs[ni.emBegin + k] - recBase); }
            if (ni.treeCount) { n.trees.resize(ni.treeCount); for (uint32_t k = 0; k < ni.treeCount; ++k) n.trees[k] = ni.treeBegin + k; }
        }
        contigs.clear();
        nodeSeq.resize(nodes.size());
        for (size_t i = 0; i < nodes.size(); ++i) { const int32_t v = nodes[i].vrec; nodeSeq[i] = NodeSeq{seq(v), (uint16_t)seqLen(v), (uint16_t)minimPos(v)}; }
    }

    void encodeLz(const Batch& batch, const Batch& graph, const BinIn& bin, const ArchiveParams& arch, BinStreams& o)
    {
        const bool stageTrace = getenv("FS_BIN_TRACE") && bin.recCount >= (atoi(getenv("FS_BIN_TRACE")) > 1 ? (uint32_t)atoi(getenv("FS_BIN_TRACE")) : 40000u);      // stage clock of the heaviest bins (design studies; FS_BIN_TRACE=1: 40 000 records and more, n: n and more)
        auto clk = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6; };
        const double t0 = stageTrace ? clk() : 0; double t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        setArchive(arch);
        B = &batch; G = &graph; out = &o; curSig = bin.signature; curBin = &bin;
        o.reset(pe ? S_PE_COUNT : S_SE_COUNT);
        if (cfg.quaParams.method == MET_QVZ) well.reset(arch.qvz.wellSeed);
        initNodes(bin);
        // (a slice's input and device-written streams share one 32-bit address space, and the streams get room for the most their ops
        // can write -- about five bytes per base; a bin whose share would pass a quarter of a gigabyte keeps the walk's own loops)
        devEmit = wantDevEmit && bin.recCount > 0 && bin.maxLen <= 255 && (uint64_t)bin.rawDnaSize * 5u <= (256ull << 20);
        if (devEmit) {      // the bin's bases: one stretch of the batch's base array (its records were unpacked one behind the other)
            uint64_t lo = ~0ull, hi = 0;
            for (uint32_t i = 0; i < bin.recCount; ++i) { const Rec& r = B->recs[bin.recBegin + i]; lo = std::min<uint64_t>(lo, r.seqOff); hi = std::max<uint64_t>(hi, (uint64_t)r.seqOff + r.seqLen + r.auxLen); }
            if (hi - lo > 0xFFFFFF00ull) devEmit = false;
            else { o.deviceEmit = true; o.emitSeqLo = lo; o.emitSeqHi = hi; o.emitSeq = B->seq.data() + lo; o.emitBound[fsdev::ECH_COUNT] = 2; }
        }
        lzStack.clear(); consStack.clear();
        lzStack.push_back(LzContext());
        matchRle.start(&o.s[S_Match]); consMatchRle.start(&o.s[S_CMatch]); lzRle0.start(&o.s[S_LzId]);
        if (pe) { matchRlePE.start(&o.s[S_MatchRlePE]); resetPair(); pairTodo.clear(); }
        // CompressRecords (FastqCompressor.cpp:1228-1276): sort the top-level nodes, match, then per root contigs + BFS
        std::vector<int32_t> order(bin.topCount);
        for (uint32_t k = 0; k < bin.topCount; ++k) order[k] = (int32_t)(G->topNodes[bin.topBegin + k] - nodeBase);
        introsort(order.data(), order.size(), [&](int32_t a, int32_t b) { return compareReads(nodes[a].vrec, nodes[b].vrec); });
        std::vector<int32_t> roots;
        if (stageTrace) t1 = clk();
        // the window searches of all the bin's constructions, on the device (matcher.hip); a row that names a dummy slot
        // (reads of mostly 'N', manual thresholds) sends the whole bin through the host scan, which keeps the
        // reference's handling of that corner
        havePre = false;
        const bool deviceSearch = (bool)matcher && !matchTrace && (!matcherGate || matcherGate());
        const bool wantTable = deviceSearch || matchTrace != nullptr;
        if (wantTable) buildMatchTable(order);
        if (matchTrace && traceResize) { matchTrace->assign(mReads.size(), fsdev::MatchRow{-1, 0, 0, 0, 0, 0, 0}); }
        if (deviceSearch && par.maxLzWindowSize >= 2 && par.maxLzWindowSize <= 1025 && bin.maxLen <= 256) {
            mRows.resize(mReads.size());
            if (matcher(B->seq.data() + matchSeqBase, matchSeqBytes, packedDna(), mReads.data(), mReads.size(), mCalls.data(), mCalls.size(), mWarm.data(), mWarm.size(), matchParams(), mRows.data())) {
                havePre = true;
                for (const fsdev::MatchCall& c : mCalls) for (uint32_t i = 0; i < c.count && havePre; ++i) if (mRows[c.first + i].dummy) havePre = false;
            }
        }
        if (stageTrace) t2 = clk();
        constructMatchTree(order, roots, -1, wantTable ? 0 : -1);
        if (stageTrace) t3 = clk();
#ifdef FS_DEBUG_DUMP
#endif
        for (int32_t root : roots) {
            if (!nodes[root].children.empty()) buildContigs(root);
            encodeTree(root, false);
        }
        matchRle.end(); consMatchRle.end(); lzRle0.end();
        const double tp = stageTrace ? clk() : 0;
        if (pe) { finishPairs(); if (!o.pairsPending) matchRlePE.end(); }
        if (stageTrace) { t4 = clk(); fprintf(stderr, "[bin] %u records: nodes + sort %.1f ms, match table + device search %.1f ms (pre %d), top-level tree %.1f ms, contigs + sub-trees + emission %.1f ms, mate searches + their streams %.1f ms\n", bin.recCount, t1 - t0, t2 - t1, (int)havePre, t3 - t2, tp - t3, t4 - tp); }
    }
};

}  // namespace fs

#include "frontend_pe.inc"

namespace fs {

BinEncoder::BinEncoder(const PackParams& par) : impl_(new Impl(par)) {}
void BinEncoder::setMatcher(MatchFn fn) { impl_->matcher = std::move(fn); }
void BinEncoder::setMatcherGate(std::function<bool()> fn) { impl_->matcherGate = std::move(fn); }
void BinEncoder::setDeviceEmit(bool on) { impl_->wantDevEmit = on; }
void BinEncoder::setAsyncMates(AsyncMateFn fn) { impl_->asyncMates = std::move(fn); }
void BinEncoder::setMateMatcher(MateFn fn) { impl_->mateMatcher = std::move(fn); }
void BinEncoder::checkMateMatcher(const Batch& data, const Batch& graph, const BinIn& bin, const ArchiveParams& arch, const MateFn& fn, uint64_t& pairs, uint64_t& differing)
{
    Impl& m = *impl_;
    BinStreams tmp;
    std::vector<fsdev::MateRow> host, dev;
    const MateFn keep = m.mateMatcher; const MatchFn keepSe = m.matcher;
    m.matcher = nullptr; m.mateMatcher = nullptr; m.mateTrace = &host;
    m.encodeLz(data, graph, bin, arch, tmp);                       // pass 1: the host search, its rows traced
    m.mateTrace = &dev; m.mateMatcher = fn;
    m.encodeLz(data, graph, bin, arch, tmp);                       // pass 2: the device's rows (an exception if it could not run)
    m.mateTrace = nullptr; m.mateMatcher = keep; m.matcher = keepSe;
    if (host.size() != dev.size()) throw std::runtime_error("mate search check: the two passes met different numbers of pairs");
    for (size_t i = 0; i < host.size(); ++i) {
        const fsdev::MateRow &a = host[i], &b = dev[i];
        ++pairs;
        if (b.overflow) throw std::runtime_error("mate search check: the device's alignment list overflowed");
        bool same = a.cost == b.cost;
        if (same && a.cost < 255) same = a.shift == b.shift && a.no_mismatches == b.no_mismatches;
        if (same && a.match >= 0) same = a.match == b.match && a.prev_id == b.prev_id;
        if (same && a.match < 0) same = b.match < 0;
        if (!same) ++differing;
    }
}
void BinEncoder::checkMatcher(const Batch& data, const Batch& graph, const BinIn& bin, const ArchiveParams& arch, const MatchFn& fn, uint64_t& reads, uint64_t& differing)
{
    Impl& m = *impl_;
    BinStreams tmp;
    std::vector<fsdev::MatchRow> host;
    const MatchFn keep = m.matcher; m.matcher = nullptr;
    // pass 1: the host scan, tracing its answers (the table is built inside; size the trace when it is known)
    m.matchTrace = &host;
    m.traceResize = true;
    m.encodeLz(data, graph, bin, arch, tmp);
    m.matchTrace = nullptr; m.traceResize = false;
    const std::vector<fsdev::MatchRead> tReads = m.mReads; const std::vector<fsdev::MatchCall> tCalls = m.mCalls; const std::vector<uint32_t> tWarm = m.mWarm;
    m.curBin = &bin; m.B = &data;
    std::vector<fsdev::MatchRow> dev(tReads.size());
    memset(dev.data(), 0, dev.size() * sizeof(fsdev::MatchRow));
    if (!fn(data.seq.data() + m.matchSeqBase, m.matchSeqBytes, m.packedDna(), tReads.data(), tReads.size(), tCalls.data(), tCalls.size(), tWarm.data(), tWarm.size(), m.matchParams(), dev.data())) throw std::runtime_error("device matcher did not run");
    for (const fsdev::MatchCall& c : tCalls)
        for (uint32_t i = 0; i < c.count; ++i) {
            const fsdev::MatchRow &a = host[c.first + i], &b = dev[c.first + i];
            ++reads;
            if (a.match != b.match || a.cost != b.cost || a.shift != b.shift || a.no_mismatches != b.no_mismatches || a.identical != b.identical || a.dummy != b.dummy) ++differing;
        }
    m.matcher = keep;
}
BinEncoder::~BinEncoder() { if (impl_) impl_->dropPairState(); delete impl_; }
void BinEncoder::encodeLz(const Batch& batch, const BinIn& bin, const ArchiveParams& arch, BinStreams& out) { impl_->encodeLz(batch, batch, bin, arch, out); }
void BinEncoder::encodeLz(const Batch& data, const Batch& graph, const BinIn& bin, const ArchiveParams& arch, BinStreams& out) { impl_->encodeLz(data, graph, bin, arch, out); }

}  // namespace fs
