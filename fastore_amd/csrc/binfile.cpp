#include "binfile.h"
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <stdexcept>
#include "bitio.h"
#include "device_types.h"

namespace fs {

void fs_advise_huge(void* p, size_t bytes) { (void)madvise(p, bytes, MADV_HUGEPAGE); }


namespace {

struct FileHeader { uint64_t footerOffset, recordsCount, blockCount, footerSize; uint8_t usesHeaderStream; uint8_t reserved[7]; };
static_assert(sizeof(FileHeader) == 40, "BinFileHeader::HeaderSize");


uint32_t bitLength(uint64_t x) { for (uint32_t i = 0; i < 32; ++i) if (x < (1ull << i)) return i; return 64; }
const uint32_t kBitsPerClass[4] = {4, 8, 16, 30};
const uint8_t kIdxToQua8[8] = {0, 6, 15, 22, 27, 33, 37, 40};

struct Settings {
    bool hasConstLen = false, hasReadGroups = false, usesHeaders = false;
    uint32_t minLen = (uint32_t)-1, maxLen = 0, suffixLen = 0, bitsPerLen = 0, signatureId = 0;
    char signature[32];
};

void generateMinimizer(const MinimizerParametersRaw& mp, uint32_t id, char* buf)
{
    const uint32_t total = 1u << (2 * mp.signatureLen);
    if (id == total) { for (int i = 0; i < mp.signatureLen; ++i) buf[i] = 'N'; return; }
    for (int i = mp.signatureLen - 1; i >= 0; --i) { buf[i] = mp.dnaSymbolOrder[id & 3]; id >>= 2; }
}

struct Unpacker {
    const BinModuleConfigRaw& cfg;
    Batch& b;        // bases / qualities / headers / records
    Batch& g;        // graph tables
    BitReader meta, dna, qua, head;
    bool pe;
    Settings pairSettings;
    bool placed = false; uint64_t seqCur = 0, headCur = 0;
    uint64_t seqEnd = 0, headEnd = 0;     // placed mode: the bin's own range of the shared arrays ends here (the next bin's begins)
    uint32_t recEnd = 0;     // one past the last record of the slice being read: group sizes in a damaged stream cannot walk past it
    uint32_t takeRec(uint32_t& recIdx) { if (recIdx >= recEnd) throw std::runtime_error("Corrupted bin: more records in the graph than in the footer"); return recIdx++; }
    void checkGroup(uint32_t groupSize, uint32_t recIdx, uint32_t slack) const { if ((uint64_t)groupSize > (uint64_t)(recEnd - recIdx) + slack) throw std::runtime_error("Corrupted bin: group larger than the bin"); }
    uint32_t dna4[256];      // byte of packed bases -> four characters

    // packedQ: the qualities stay packed (Batch::quaPacked / quaBit): readQuality only notes where a string starts
    bool packedQ = false; uint64_t quaStart = 0; uint32_t quaBits = 6;      // bits per stored score: 6 (lossless), 3 (8-bin), 1 (binary)
    Unpacker(const BinModuleConfigRaw& c, Batch& batch, Batch& graph, const std::vector<uint8_t>& m, uint64_t ms, const std::vector<uint8_t>& d, uint64_t ds,
             const uint8_t* q, uint64_t qs, const uint8_t* h, uint64_t hs)
        : cfg(c), b(batch), g(graph), meta(m.data(), ms), dna(d.data(), ds), qua(q, qs), head(h, hs), pe(c.archiveType.readType == READ_PE)
    {
        pairSettings.minLen = pairSettings.maxLen = 1; pairSettings.hasConstLen = true; pairSettings.usesHeaders = false;
        const char* o = c.minimizer.dnaSymbolOrder;
        for (uint32_t w = 0; w < 256; ++w) { const uint8_t q[4] = {(uint8_t)o[w >> 6], (uint8_t)o[(w >> 4) & 3], (uint8_t)o[(w >> 2) & 3], (uint8_t)o[w & 3]}; memcpy(&dna4[w], q, 4); }
    }

    // packedD: the packed bases are kept beside the unpacked ones (Batch::dnaPacked / dnaBit / dnaInfo): where the last readDna
    // began in the bin's bytes, and in which form it was stored
    bool packedD = false; uint64_t dnaBitBase = 0, dnaStart = 0; bool dnaPlain = false, dnaTooLong = false;      // dnaTooLong: a bit offset past 32 bits -- the bin's packed form is not offered (ASCII bases go up)
    void readDna(uint8_t* seq, uint32_t seqLen, uint32_t minimPos, uint32_t suffixLen)
    {
        const bool plain = meta.getBit() != 0;
        dnaStart = dnaBitBase + dna.bitPosition(); dnaPlain = plain;
        const char* idxToDna = cfg.minimizer.dnaSymbolOrder;
        if (plain) {
            // sixteen bases per window read: four table look-ups of four bases each
            auto run = [&](uint32_t from, uint32_t to) {
                uint32_t i = from;
                for (; i + 16 <= to; i += 16) {
                    const uint32_t w = dna.getBits(32);
                    memcpy(seq + i, &dna4[w >> 24], 4); memcpy(seq + i + 4, &dna4[(w >> 16) & 255], 4); memcpy(seq + i + 8, &dna4[(w >> 8) & 255], 4); memcpy(seq + i + 12, &dna4[w & 255], 4);
                }
                for (; i + 4 <= to; i += 4) { const uint32_t w = dna.getBits(8); memcpy(seq + i, &dna4[w], 4); }
                for (; i < to; ++i) seq[i] = (uint8_t)idxToDna[dna.get2Bits()];
            };
            run(0, minimPos); run(minimPos + suffixLen, seqLen);
        } else {
            // three bits per base: codes 5-7 do not exist (dnaSymbolOrder has five entries)
            auto base3 = [&]() -> uint8_t { const uint32_t c = dna.getBits(3); if (c > 4) throw std::runtime_error("Corrupted bin: invalid base code"); return (uint8_t)idxToDna[c]; };
            for (uint32_t i = 0; i < minimPos; ++i) seq[i] = base3();
            for (uint32_t i = minimPos + suffixLen; i < seqLen; ++i) seq[i] = base3();
        }
    }
    void readQuality(uint64_t at, uint32_t n)
    {
        if (packedQ) { quaStart = qua.bitPosition(); qua.skipBits((uint64_t)quaBits * n); return; }
        uint8_t* q = b.qua.data() + at;
        const uint32_t off = cfg.archiveType.qualityOffset;
        switch (cfg.quaParams.method) {
        case MET_BINARY: for (uint32_t i = 0; i < n; ++i) q[i] = (uint8_t)(off + (qua.getBit() ? 40 : 6)); break;
        case MET_8BIN: for (uint32_t i = 0; i < n; ++i) q[i] = (uint8_t)(off + kIdxToQua8[qua.getBits(3)]); break;
        default: qua.unpack6(q, n, off); break;
        }
    }
    bool packedH = false;      // the read ids stay packed (Batch::headPacked / headBit): readHeader only notes where the characters start
    void readHeader(Rec& r)
    {
        r.headLen = (uint8_t)head.getBits(8);
        if (packedH) {
            const uint64_t at = head.bitPosition();
            if (at > 0xFFFFFFFFull) throw std::runtime_error("bin with more than 512 MiB of packed read ids");
            b.headBit[(size_t)(&r - b.recs.data())] = (uint32_t)at; r.headOff = 0;
            if (r.headLen > 1) head.skipBits(7ull * (r.headLen - 1u));
            return;
        }
        if (placed) { if (headCur + r.headLen > headEnd) throw std::runtime_error("bin footer understates the header bytes"); r.headOff = (uint32_t)headCur; headCur += r.headLen; }
        else { r.headOff = (uint32_t)b.head.size(); b.head.resize(b.head.size() + r.headLen); }
        uint8_t* h = b.head.data() + r.headOff;
        if (r.headLen) h[0] = '@';
        uint32_t i = 1;
        for (; i + 4 <= r.headLen; i += 4) { const uint32_t w = head.getBits(28); h[i] = (uint8_t)(w >> 21); h[i + 1] = (uint8_t)((w >> 14) & 127); h[i + 2] = (uint8_t)((w >> 7) & 127); h[i + 3] = (uint8_t)(w & 127); }
        for (; i < r.headLen; ++i) h[i] = (uint8_t)head.getBits(7);
    }
    // IFastqPacker::ReadNextRecord on the bytes [seq, seq+len): returns false when the dna stream is exhausted
    bool readNextRecord(const Settings& s, Rec& r, uint32_t seqOff, uint32_t len, bool isMate2)
    {
        if (dna.position() >= dna.size()) return false;
        uint32_t minimPos = 0;
        if (s.suffixLen != 0) {
            const bool rev = meta.getBit() != 0;
            if (rev) r.flags |= FLAG_REVERSE; else r.flags &= ~FLAG_REVERSE;
            r.minimPos = (uint16_t)meta.getBits(8);
            minimPos = r.minimPos;
        } else if (!isMate2) { r.flags &= ~FLAG_REVERSE; r.minimPos = 0; }
        // before any base is stored: the position comes from 8 untrusted bits
        if (s.suffixLen != 0 && minimPos + s.suffixLen > len) throw std::runtime_error("Corrupted bin: signature position outside the read");
        readDna(b.seq.data() + seqOff, len, minimPos, s.suffixLen);
        if (packedD && !isMate2) {      // (the window search reads first mates only)
            const size_t idx = (size_t)(&r - b.recs.data());
            if (dnaStart > 0xFFFFFFFFull) dnaTooLong = true;
            b.dnaBit[idx] = (uint32_t)dnaStart;
            b.dnaInfo[idx] = (dnaPlain ? fsdev::PACKED_PLAIN : 0u) | (s.suffixLen != 0 ? fsdev::PACKED_HAS_SIG | (s.signatureId & ((1u << fsdev::PACKED_SIG_BITS) - 1u)) | (minimPos << fsdev::PACKED_SIG_BITS) : 0u);
        }
        readQuality(seqOff, len);
        if (packedQ) noteQuality(r, len, isMate2);
        if (s.usesHeaders) readHeader(r);
        return true;
    }
    // the record's bit offset (first mate); a second mate's scores must follow the first's directly
    void noteQuality(const Rec& r, uint32_t len, bool isMate2)
    {
        const size_t idx = (size_t)(&r - b.recs.data());
        if (quaStart > 0xFFFFFFFFull) throw std::runtime_error("bin with more than 512 MiB of packed qualities");
        if (!isMate2) b.quaBit[idx] = (uint32_t)quaStart;
        else if (quaStart != (uint64_t)b.quaBit[idx] + (uint64_t)quaBits * r.seqLen) throw std::runtime_error("Corrupted bin: mate qualities are not adjacent");
        (void)len;
    }
    uint32_t allocSeq(uint32_t n)
    {
        if (placed) { if (seqCur + n > seqEnd) throw std::runtime_error("bin footer understates the bases"); const uint64_t o = seqCur; seqCur += n; return (uint32_t)o; }
        const uint64_t off = b.seq.size();
        if (off + n > 0xFFFFFFF0ull) throw std::runtime_error("batch exceeds 4 GiB of bases");
        b.seq.resize(off + n); if (!packedQ) b.qua.resize(off + n);
        return (uint32_t)off;
    }
    void readRecordData(const Settings& s, Rec& r)
    {
        r.flags = 0; r.minimPos = 0; r.headLen = 0; r.headOff = (uint32_t)(placed ? headCur : b.head.size());
        if (s.hasConstLen) { r.seqLen = (uint16_t)s.minLen; r.auxLen = pe ? r.seqLen : 0; }
        else {
            r.seqLen = (uint16_t)(meta.getBits(s.bitsPerLen) + s.minLen);
            r.auxLen = pe ? (uint16_t)(meta.getBits(s.bitsPerLen) + s.minLen) : 0;
        }
        if (pe && s.suffixLen) { if (meta.getBit()) r.flags |= FLAG_SWAPPED; }
        r.seqOff = allocSeq((uint32_t)r.seqLen + r.auxLen);
        readNextRecord(s, r, r.seqOff, r.seqLen, false);
        if (s.suffixLen > 0) {
            if ((uint32_t)r.minimPos + cfg.minimizer.signatureLen > r.seqLen) throw std::runtime_error("Corrupted bin: signature position outside the read");
            memcpy(b.seq.data() + r.seqOff + r.minimPos, s.signature, cfg.minimizer.signatureLen);
        }
        if (pe) readNextRecord(pairSettings, r, r.seqOff + r.seqLen, r.auxLen, true);
    }
    void readExactMatch(const Settings& s, const Rec& mainRec, Rec& r)
    {
        r.flags = 0; r.headLen = 0; r.headOff = (uint32_t)(placed ? headCur : b.head.size());
        if (meta.getBit()) r.flags |= FLAG_REVERSE;
        if (pe && meta.getBit()) r.flags |= FLAG_SWAPPED;
        r.seqLen = mainRec.seqLen; r.auxLen = pe ? mainRec.auxLen : 0;
        r.seqOff = allocSeq((uint32_t)r.seqLen + r.auxLen);
        memcpy(b.seq.data() + r.seqOff, b.seq.data() + mainRec.seqOff, mainRec.seqLen);
        r.minimPos = mainRec.minimPos;
        if (packedD) { const size_t idx = (size_t)(&r - b.recs.data()), mi = (size_t)(&mainRec - b.recs.data()); b.dnaBit[idx] = b.dnaBit[mi]; b.dnaInfo[idx] = b.dnaInfo[mi]; }
        readQuality(r.seqOff, r.seqLen);
        if (packedQ) noteQuality(r, r.seqLen, false);
        if (cfg.archiveType.readsHaveHeaders) readHeader(r);
        if (pe) readNextRecord(pairSettings, r, r.seqOff + r.seqLen, r.auxLen, true);
    }
    // IFastqNodesPacker::ReadNextNode -- node slot `nodeIdx` must already exist
    void readNextNode(uint32_t nodeIdx, const Settings& s, uint32_t& recIdx)
    {
        const uint32_t mainRec = takeRec(recIdx);
        readRecordData(s, b.recs[mainRec]);
        g.nodes[nodeIdx].rec = mainRec;
        if (!s.hasReadGroups) return;
        const bool hasEm = meta.getBit() != 0;
        const bool hasTrees = meta.getBit() != 0;
        if (hasEm) {
            const uint32_t groupSize = meta.getBits(kBitsPerClass[meta.get2Bits()]);
            checkGroup(groupSize, recIdx, 0);
            g.nodes[nodeIdx].emBegin = (uint32_t)g.emRecs.size();
            g.nodes[nodeIdx].emCount = groupSize;
            g.emRecs.resize(g.emRecs.size() + groupSize);
            for (uint32_t i = 0; i < groupSize; ++i) {
                const uint32_t em = takeRec(recIdx);
                readExactMatch(s, b.recs[mainRec], b.recs[em]);
                g.emRecs[g.nodes[nodeIdx].emBegin + i] = em;
            }
        }
        if (hasTrees) {
            const uint32_t tCount = meta.getBits(kBitsPerClass[meta.get2Bits()]);
            checkGroup(tCount, recIdx, 16);
            const uint32_t treeBegin = (uint32_t)g.trees.size();
            g.nodes[nodeIdx].treeBegin = treeBegin; g.nodes[nodeIdx].treeCount = tCount;
            g.trees.resize(g.trees.size() + tCount);
            for (uint32_t t = 0; t < tCount; ++t) {
                TreeIn tr;
                tr.signatureId = meta.getBits(cfg.minimizer.signatureLen * 2);
                tr.mainSignaturePos = (int32_t)meta.getBits(8);
                const uint32_t groupSize = meta.getBits(kBitsPerClass[meta.get2Bits()]);
                checkGroup(groupSize, recIdx, 0);
                tr.nodeBegin = (uint32_t)g.nodes.size(); tr.nodeCount = groupSize;
                g.nodes.resize(g.nodes.size() + groupSize, NodeIn{0, 0, 0, 0, 0});
                g.trees[treeBegin + t] = tr;
                Settings cs = s;
                cs.signatureId = tr.signatureId; cs.suffixLen = cfg.minimizer.signatureLen;
                generateMinimizer(cfg.minimizer, cs.signatureId, cs.signature);
                for (uint32_t i = 0; i < groupSize; ++i) readNextNode(tr.nodeBegin + i, cs, recIdx);
            }
        }
    }
};

}  // namespace

BinFile::~BinFile() { close(); }

void BinFile::close() { unmap(meta_); unmap(dna_); unmap(qua_); unmap(headf_); }

// The four stream files are mapped read-only: a signature's slices are scattered over them in
// thousands of small pieces (one per bin/rebin worker flush), which made seek+read the dominant cost.
BinFile::Map BinFile::mapFile(const std::string& name)
{
    const int fd = ::open(name.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("Cannot open file to read: " + name);      // (FileStreamReader, fastore_bin/FileStream.cpp:134)
    struct stat st;
    if (fstat(fd, &st) != 0) { ::close(fd); throw std::runtime_error("Cannot open file: " + name); }
    Map m; m.size = (uint64_t)st.st_size;
    if (m.size) {
        void* p = mmap(nullptr, m.size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (p == MAP_FAILED) { ::close(fd); throw std::runtime_error("Cannot open file: " + name); }
        m.p = (const uint8_t*)p;
    }
    ::close(fd);
    return m;
}
void BinFile::unmap(Map& m) { if (m.p) munmap((void*)m.p, m.size); m.p = nullptr; m.size = 0; }
void BinFile::copyAt(const Map& m, uint64_t off, void* dst, uint64_t n, const char* what)
{
    if (n == 0) return;
    if (off > m.size || n > m.size - off) throw std::runtime_error(std::string("Cannot read ") + what);      // (off + n may wrap)
    memcpy(dst, m.p + off, n);
}

void BinFile::open(const std::string& prefix, uint32_t minBinSize)
{
    close();
    meta_ = mapFile(prefix + ".bmeta");
    const uint64_t metaSize = meta_.size;
    if (metaSize == 0) throw std::runtime_error("Empty file.");
    dna_ = mapFile(prefix + ".bdna");
    qua_ = mapFile(prefix + ".bqua");
    FileHeader fh; memset(&fh, 0, sizeof fh);
    copyAt(meta_, 0, &fh, sizeof fh, "bin header");
    if (fh.blockCount == 0 || fh.footerOffset > metaSize || fh.footerSize > metaSize - fh.footerOffset) throw std::runtime_error("Corrupted archive header");
    usesHeaderStream_ = fh.usesHeaderStream != 0;
    if (usesHeaderStream_) headf_ = mapFile(prefix + ".bhead");
    std::vector<uint8_t> footer(fh.footerSize);
    copyAt(meta_, fh.footerOffset, footer.data(), fh.footerSize, "bin footer");
    readFooter(footer);
    // BinFileExtractor::StartDecompress: split signatures (N bin excluded) by record count
    std_.clear(); small_.clear();
    const uint32_t nSig = nSignature();
    for (const auto& kv : bins_) {
        if (kv.first == nSig) continue;
        if (kv.second.totalRecordsCount >= minBinSize) std_.push_back(kv.first); else small_.push_back(kv.first);
    }
}

void BinFile::readFooter(const std::vector<uint8_t>& buf)
{
    BitReader r(buf.data(), buf.size());
    r.getBytes(&cfg_, sizeof cfg_);
    {   // the raw config steers table sizes and the base alphabet: refuse values the writers never produce
        const MinimizerParametersRaw& mp = cfg_.minimizer;
        bool ok = mp.signatureLen >= 4 && mp.signatureLen <= 12 && mp.signatureMaskCutoffBits <= 2 * mp.signatureLen && cfg_.quaParams.method <= MET_QVZ && cfg_.archiveType.readType <= READ_PE;
        uint32_t seen = 0;
        for (char c : mp.dnaSymbolOrder) { const char* at = strchr("ACGTN", c); if (!c || !at) { ok = false; break; } seen |= 1u << (at - "ACGTN"); }
        if (!ok || seen != 31u) throw std::runtime_error("Corrupted archive header");
    }
    const uint32_t total = (1u << (2 * cfg_.minimizer.signatureLen)) + 1;
    std::vector<bool> bitmap(total);
    for (uint32_t i = 0; i < total; ++i) bitmap[i] = r.getBit() != 0;
    r.flushWord();
    bins_.clear();
    for (uint32_t i = 0; i < total; ++i) {
        if (!bitmap[i]) continue;
        BinInfo& bi = bins_[i];
        bi.totalMetaSize = r.get8Bytes(); bi.totalDnaSize = r.get8Bytes(); bi.totalQuaSize = r.get8Bytes();
        bi.totalRawDnaSize = r.get8Bytes(); bi.totalRecordsCount = r.get8Bytes();
        if (usesHeaderStream_) { bi.totalHeadSize = r.get8Bytes(); bi.totalRawHeadSize = r.get8Bytes(); }
        const uint64_t n = r.get8Bytes();
        if (n > (buf.size() - std::min<uint64_t>(buf.size(), r.position())) / sizeof(BlockMetaDataRaw)) throw std::runtime_error("Corrupted archive header");
        bi.blocks.resize(n);
        r.getBytes(bi.blocks.data(), n * sizeof(BlockMetaDataRaw));
        // The totals size the unpack buffers: hold them against what the stream files can hold (every base has a
        // quality of at least one bit -- exact duplicates store no bases, but they do store qualities; positions under an
        // 'N' may be skipped, and fewer than a third of a read is 'N'; a record costs at least one bit; 7 bits per read-id
        // character), so that a damaged footer is an error, not a terabyte allocation or a copy past a buffer.
        const uint64_t sigLen = cfg_.minimizer.signatureLen;
        auto plausible = [&](uint64_t meta, uint64_t dna, uint64_t qua, uint64_t head, uint64_t rawDna, uint64_t recs, uint64_t rawHead) {
            if (meta > meta_.size || dna > dna_.size || qua > qua_.size || head > headf_.size) return false;
            if (recs > 8 * (meta + dna) + 64) return false;
            if (rawDna > 12 * qua + (sigLen + 64) * recs + 64) return false;
            if (rawHead > 2 * head + 64) return false;
            return true;
        };
        uint64_t sm = 0, sd = 0, sq = 0, sh = 0, sr = 0, sn = 0, srh = 0;
        for (const BlockMetaDataRaw& b : bi.blocks) {
            if (!plausible(b.metaSize, b.dnaSize, b.quaSize, usesHeaderStream_ ? b.headSize : 0, b.rawDnaSize, b.recordsCount, usesHeaderStream_ ? b.rawHeadSize : 0)) throw std::runtime_error("Corrupted archive header");
            sm += b.metaSize; sd += b.dnaSize; sq += b.quaSize; sh += b.headSize; sr += b.rawDnaSize; sn += b.recordsCount; srh += b.rawHeadSize;
        }
        if (!plausible(bi.totalMetaSize, bi.totalDnaSize, bi.totalQuaSize, bi.totalHeadSize, bi.totalRawDnaSize, bi.totalRecordsCount, bi.totalRawHeadSize)
            || sm > bi.totalMetaSize || sd > bi.totalDnaSize || sq > bi.totalQuaSize || (usesHeaderStream_ && sh > bi.totalHeadSize)
            || sr > bi.totalRawDnaSize || sn > bi.totalRecordsCount || (usesHeaderStream_ && srh > bi.totalRawHeadSize))
            throw std::runtime_error("Corrupted archive header");
    }
    qvz_ = QvzModel();
    if (cfg_.quaParams.method == MET_QVZ) qvz_.parse(r);      // WELL seed, max read length, codebook (BinFile.cpp:740-755)
    head_ = HeaderStats();
    if (usesHeaderStream_) {
        const uint32_t fields = r.getByte();
        head_.fields.resize(fields);
        for (HeaderField& f : head_.fields) {
            f.isNumeric = r.getByte() != 0;
            f.isConst = r.getByte() != 0;
            f.separator = (char)r.getByte();
            if (f.isNumeric) {
                f.minValue = r.get8Bytes();
                if (!f.isConst) f.maxValue = r.get8Bytes();
            } else {
                uint32_t possible = 1;
                if (!f.isConst) possible = r.get2Bytes();
                std::set<std::string> vals;
                for (uint32_t i = 0; i < possible; ++i) {
                    const uint32_t ss = r.getByte();
                    std::string s(ss, '\0');
                    r.getBytes(&s[0], ss);
                    vals.insert(s);
                }
                f.possibleValues.assign(vals.begin(), vals.end());
            }
        }
        if (cfg_.archiveType.readType == READ_PE) head_.pairedEndFieldIdx = r.getByte();
    }
}

void BinFile::unpack(uint32_t signature, Batch& batch, bool asNewBin, bool keepPackedDna) const
{
    int64_t at = -1;
    if (keepPackedDna && asNewBin) { at = (int64_t)((batch.dnaPacked.size() + 15u) & ~(size_t)15u); batch.dnaPacked.resize((size_t)at + bins_.at(signature).totalDnaSize + 16u); }
    unpackImpl(signature, batch, batch, asNewBin, false, 0, 0, 0, -1, -1, at);
}
void BinFile::unpackPlaced(uint32_t signature, Batch& data, uint64_t seqBase, uint64_t headBase, uint32_t recBase, Batch& graph, int64_t quaBase, int64_t headPackedBase, int64_t dnaPackedBase) const
{ unpackImpl(signature, data, graph, true, true, seqBase, headBase, recBase, quaBase, headPackedBase, dnaPackedBase); }

void BinFile::unpackImpl(uint32_t signature, Batch& data, Batch& graph, bool asNewBin, bool placed, uint64_t seqBase, uint64_t headBase, uint32_t recBase, int64_t quaBase, int64_t headPackedBase, int64_t dnaPackedBase) const
{
    // gather buffers of the calling thread, kept across bins: fresh vectors of this size are mmap'ed by malloc, and the
    // map/unmap/first-touch churn of thousands of them per second serialises the host threads in the kernel
    static thread_local std::vector<uint8_t> bMeta_, bDna_, bQua_, bHead_;
    const auto it = bins_.find(signature);
    if (it == bins_.end()) throw std::runtime_error("signature not present in bin file");
    const BinInfo& bi = it->second;
    // BinFileReader::ReadBlock: gather the signature's slices from the four streams
    if (bMeta_.size() < bi.totalMetaSize) bMeta_.resize(bi.totalMetaSize);
    if (bDna_.size() < bi.totalDnaSize) bDna_.resize(bi.totalDnaSize);
    const bool packedD = dnaPackedBase >= 0 && asNewBin;
    if (packedD && (uint64_t)dnaPackedBase + bi.totalDnaSize > data.dnaPacked.size()) throw std::runtime_error("packed bases: batch arrays not sized");
    const bool packedQ = quaBase >= 0;
    if (packedQ) {
        if (!placed) throw std::runtime_error("packed qualities: placed unpack only");
        if ((uint64_t)quaBase + bi.totalQuaSize > data.quaPacked.size() || data.quaBit.size() != data.recs.size()) throw std::runtime_error("packed qualities: batch arrays not sized");
    } else if (bQua_.size() < bi.totalQuaSize) bQua_.resize(bi.totalQuaSize);
    uint8_t* const quaDst = packedQ ? data.quaPacked.data() + quaBase : bQua_.data();
    const bool packedH = headPackedBase >= 0 && usesHeaderStream_;
    if (packedH) {
        if (!placed || (uint64_t)headPackedBase + bi.totalHeadSize > data.headPacked.size() || data.headBit.size() != data.recs.size()) throw std::runtime_error("packed read ids: batch arrays not sized");
    } else if (usesHeaderStream_ && bHead_.size() < bi.totalHeadSize) bHead_.resize(bi.totalHeadSize);
    uint8_t* const headDst = packedH ? data.headPacked.data() + headPackedBase : bHead_.data();
    uint64_t mo = 0, dO = 0, qo = 0, ho = 0, rawDna = 0, records = 0;
    for (const BlockMetaDataRaw& blk : bi.blocks) {
        copyAt(meta_, blk.metaFileOffset, bMeta_.data() + mo, blk.metaSize, ".bmeta"); mo += blk.metaSize;
        copyAt(dna_, blk.dnaFileOffset, bDna_.data() + dO, blk.dnaSize, ".bdna"); dO += blk.dnaSize;
        copyAt(qua_, blk.quaFileOffset, quaDst + qo, blk.quaSize, ".bqua"); qo += blk.quaSize;
        if (usesHeaderStream_) { copyAt(headf_, blk.headFileOffset, headDst + ho, blk.headSize, ".bhead"); ho += blk.headSize; }
        rawDna += blk.rawDnaSize; records += blk.recordsCount;
    }
    if (asNewBin || graph.bins.empty()) {
        BinIn nb{}; nb.signature = signature; nb.topBegin = (uint32_t)graph.topNodes.size();
        nb.recBegin = placed ? recBase : (uint32_t)data.recs.size();
        graph.bins.push_back(nb);
    }
    BinIn& bin = graph.bins.back();
    bin.rawDnaSize += rawDna;
    if (packedD) {      // (the reader below works on the gather buffer: the kept copy is the device's)
        memcpy(data.dnaPacked.data() + dnaPackedBase, bDna_.data(), dO);
        bin.dnaPackedOff = (uint64_t)dnaPackedBase; bin.dnaPackedBytes = dO;
    }
    uint32_t recIdx;
    if (placed) { recIdx = recBase; if ((uint64_t)recBase + records > data.recs.size()) throw std::runtime_error("bin footer understates the records"); }
    else { recIdx = (uint32_t)data.recs.size(); data.recs.resize(data.recs.size() + records, Rec{}); }
    if (packedD) {
        if (!placed) { data.dnaBit.resize(data.recs.size(), 0u); data.dnaInfo.resize(data.recs.size(), 0u); }
        else if (data.dnaBit.size() != data.recs.size() || data.dnaInfo.size() != data.recs.size()) throw std::runtime_error("packed bases: batch arrays not sized");
    }
    const uint32_t recFirst = recIdx;

    Unpacker u(cfg_, data, graph, bMeta_, mo, bDna_, dO, quaDst, qo, headDst, ho);
    u.packedH = packedH; u.packedD = packedD;
    u.packedQ = packedQ; u.quaBits = cfg_.quaParams.method == MET_BINARY ? 1u : (cfg_.quaParams.method == MET_8BIN ? 3u : 6u);
    u.placed = placed; u.seqCur = seqBase; u.headCur = headBase;
    // a placed bin owns exactly the footer's totals of the shared arrays; a footer that understates them must not spill
    // into the neighbouring bin (another host thread is filling it)
    u.seqEnd = std::min<uint64_t>(data.seq.size(), seqBase + bi.totalRawDnaSize); u.headEnd = std::min<uint64_t>(data.head.size(), headBase + bi.totalRawHeadSize);
    Settings s;
    s.signatureId = signature;
    if (signature != nSignature()) { s.suffixLen = cfg_.minimizer.signatureLen; generateMinimizer(cfg_.minimizer, signature, s.signature); }
    s.usesHeaders = cfg_.archiveType.readsHaveHeaders != 0;
    for (const BlockMetaDataRaw& blk : bi.blocks) {
        s.minLen = u.meta.getBits(8); s.maxLen = u.meta.getBits(8);
        s.hasReadGroups = u.meta.getBit() != 0;
        s.hasConstLen = (s.minLen == s.maxLen);
        if (!s.hasConstLen) s.bitsPerLen = bitLength(s.maxLen - s.minLen);
        const uint32_t end = recIdx + (uint32_t)blk.recordsCount;
        u.recEnd = end;
        while (recIdx < end) {
            const uint32_t nodeIdx = (uint32_t)graph.nodes.size();
            graph.nodes.push_back(NodeIn{0, 0, 0, 0, 0});
            graph.topNodes.push_back(nodeIdx);
            u.readNextNode(nodeIdx, s, recIdx);
        }
        u.meta.flushWord(); u.dna.flushWord(); u.qua.flushWord(); u.head.flushWord();
    }
    if (u.dnaTooLong) { bin.dnaPackedBytes = 0; }        // (more than 512 MiB of packed bases in one bin: the window search gets its bases as ASCII)
    bin.minLen = s.minLen; bin.maxLen = s.maxLen;     // the reference keeps the last slice's values (NodesPacker.cpp:560-563)
    bin.recCount = (placed ? recIdx : (uint32_t)data.recs.size()) - bin.recBegin;
    bin.topCount = (uint32_t)graph.topNodes.size() - bin.topBegin;
    (void)recFirst;
}

}  // namespace fs
