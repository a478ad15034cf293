#include "packer.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <map>
#include <memory>
#include <chrono>
#include <stdexcept>
#include <thread>
#include <errno.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include "bitio.h"
#include "hostcoders.h"

namespace fs {

namespace {
double nowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F> void parallelFor(uint32_t n, uint32_t threads, F f)
{
    if (threads <= 1 || n <= 1) { for (uint32_t i = 0; i < n; ++i) f(i, 0u); return; }
    std::atomic<uint32_t> next(0);
    std::vector<std::thread> pool;
    std::vector<std::string> errors(threads);
    const uint32_t t = std::min(threads, n);
    for (uint32_t k = 0; k < t; ++k)
        pool.emplace_back([&, k]() {
            try { for (;;) { const uint32_t i = next.fetch_add(1); if (i >= n) break; f(i, k); } }
            catch (const std::exception& e) { errors[k] = e.what(); next.store(n); }
        });
    for (auto& th : pool) th.join();
    for (auto& e : errors) if (!e.empty()) throw std::runtime_error(e);
}
}  // namespace

// ------------------------------------------------------------------------------------------------
void Batch::append(const Batch& o)
{
    const uint64_t seqBase = seq.size(), headBase = head.size();
    const uint32_t recBase = (uint32_t)recs.size(), nodeBase = (uint32_t)nodes.size(), topBase = (uint32_t)topNodes.size(),
                   emBase = (uint32_t)emRecs.size(), treeBase = (uint32_t)trees.size();
    if (seqBase + o.seq.size() > 0xFFFFFFF0ull || headBase + o.head.size() > 0xFFFFFFF0ull) throw std::runtime_error("batch exceeds 4 GiB");
    seq.insert(seq.end(), o.seq.begin(), o.seq.end()); qua.insert(qua.end(), o.qua.begin(), o.qua.end());
    head.insert(head.end(), o.head.begin(), o.head.end());
    for (Rec r : o.recs) { r.seqOff += (uint32_t)seqBase; r.headOff += (uint32_t)headBase; recs.push_back(r); }
    for (NodeIn n : o.nodes) { n.rec += recBase; n.emBegin += emBase; n.treeBegin += treeBase; nodes.push_back(n); }
    for (uint32_t t : o.topNodes) topNodes.push_back(t + nodeBase);
    for (uint32_t e : o.emRecs) emRecs.push_back(e + recBase);
    for (TreeIn t : o.trees) { t.nodeBegin += nodeBase; trees.push_back(t); }
    for (BinIn b : o.bins) { b.recBegin += recBase; b.topBegin += topBase; bins.push_back(b); }
}

// ------------------------------------------------------------------------------------------------
void parseHeaderFields(const uint8_t* p, size_t n, bool pairedEnd, HeaderStats& out)
{
    out = HeaderStats();
    if (!p || n == 0) return;
    BitReader r(p, n);
    const uint32_t fields = r.getByte();
    out.fields.resize(fields);
    for (HeaderField& f : out.fields) {
        f.isNumeric = r.getByte() != 0; f.isConst = r.getByte() != 0; f.separator = (char)r.getByte();
        if (f.isNumeric) { f.minValue = r.get8Bytes(); if (!f.isConst) f.maxValue = r.get8Bytes(); }
        else {
            uint32_t possible = 1;
            if (!f.isConst) possible = r.get2Bytes();
            std::set<std::string> vals;
            for (uint32_t i = 0; i < possible; ++i) { const uint32_t ss = r.getByte(); std::string s(ss, '\0'); r.getBytes(&s[0], ss); vals.insert(s); }
            f.possibleValues.assign(vals.begin(), vals.end());
        }
    }
    if (pairedEnd) out.pairedEndFieldIdx = r.getByte();
}

void serializeHeaderFields(const HeaderStats& head, bool pairedEnd, std::vector<uint8_t>& out)
{
    ByteWriter w;
    w.put((uint32_t)head.fields.size());
    for (const HeaderField& f : head.fields) {
        w.put(f.isNumeric); w.put(f.isConst); w.put((uint8_t)f.separator);
        if (f.isNumeric) { w.put8(f.minValue); if (!f.isConst) w.put8(f.maxValue); }
        else {
            if (!f.isConst) w.put2((uint32_t)f.possibleValues.size());
            for (const std::string& s : f.possibleValues) { w.put((uint32_t)s.size()); w.putBytes(s.data(), s.size()); }
        }
    }
    if (pairedEnd) w.put(head.pairedEndFieldIdx);
    out = std::move(w.b);
}

// ------------------------------------------------------------------------------------------------
ArchiveWriter::~ArchiveWriter() { dropAhead(); if (meta_) fclose(meta_); if (data_) fclose(data_); }
ArchiveWriter::ArchiveWriter(ArchiveWriter&& o)
    : sizeStats_(o.sizeStats_), prefix_(std::move(o.prefix_)), meta_(o.meta_), data_(o.data_), inMemory_(o.inMemory_), mem_(std::move(o.mem_)), held_(std::move(o.held_)),
      conf_(o.conf_), sizes_(std::move(o.sizes_)), sigs_(std::move(o.sigs_)), dataBytes_(o.dataBytes_)
{ o.dropAhead(); o.meta_ = nullptr; o.data_ = nullptr; }

void ArchiveWriter::reserveAhead(uint64_t bytes)
{
    if (inMemory_ || !data_ || pre_ || (bytes < (8u << 20) && !getenv("FS_AHEAD_BYTES"))) return;
    const int fd = fileno(data_);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (uint64_t)sb.st_size >= bytes) return;
    if (fallocate(fd, 0, 0, (off_t)bytes) != 0) return;           // (file systems without it: the blocks go the usual way)
    void* mp = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (mp == MAP_FAILED) { (void)!ftruncate(fd, sb.st_size); return; }
    if (getenv("FS_TRACE")) fprintf(stderr, "[trace] archive: %llu bytes reserved ahead\n", (unsigned long long)bytes);
    pre_ = (uint8_t*)mp; preLen_ = bytes;
    preTh_ = std::thread([p = pre_, n = preLen_]() {
        // a write access that changes nothing (an atomic OR of 0): block 0 may be going into the same pages through write()
        const long pg = sysconf(_SC_PAGESIZE);
        for (size_t o = 0; o < n; o += (size_t)pg) __atomic_fetch_or(p + o, (uint8_t)0, __ATOMIC_RELAXED);
    });
}

void ArchiveWriter::dropAhead()
{
    if (preTh_.joinable()) preTh_.join();
    if (pre_) { munmap(pre_, preLen_); pre_ = nullptr; preLen_ = 0; }
}

void StreamSizeStats::start(const ArchiveTypeRaw& type, const MinimizerParametersRaw& mp)
{
    nStreams_ = type.readType == READ_PE ? 23u : 15u; hasHeaders_ = type.readsHaveHeaders != 0;
    rawSignature_ = 1u << (2 * mp.signatureLen);
    streamComp_.clear(); haveRaw_ = false; memset(rawComp_, 0, sizeof rawComp_);
}

void StreamSizeStats::addBlock(const uint8_t* data, uint64_t size, uint32_t signature)
{
    // big-endian u64s behind the 34 fixed bytes [+ raw id size]
    auto be8 = [&](uint64_t off) { uint64_t v = 0; for (int i = 0; i < 8; ++i) v = (v << 8) | data[off + i]; return v; };
    const uint64_t base = 34 + (hasHeaders_ ? 8 : 0);
    if (signature == rawSignature_) {
        if (size >= 74) { haveRaw_ = true; rawComp_[0] = be8(base); rawComp_[1] = be8(base + 8); if (hasHeaders_) { rawComp_[2] = be8(base + 16); rawComp_[3] = be8(base + 24); } }
    } else if (size >= base + 16ull * nStreams_) {
        if (streamComp_.size() < nStreams_) streamComp_.resize(nStreams_, 0);
        for (uint32_t i = 0; i < nStreams_; ++i) streamComp_[i] += be8(base + 8ull * nStreams_ + 8ull * i);
    }
}

void StreamSizeStats::print(FILE* to) const
{
    // FastqWorkBuffersSE/PE::GetBufferNames (fastore_pack/CompressedBlockData.h:129-168)
    static const char* const names[] = {"Flag", "LettersX", "Rev", "HardReads", "LzId", "Shift", "Match", "MatchBinary", "TreeShift", "CMatch", "CShift",
                                        "CLetters", "Quality", "ReadIdToken", "ReadIdValue", "PE_Flag", "PE_LettersX", "PE_Swap", "PE_Hard", "PE_LzId",
                                        "PE_Shift", "PE_MatchRLE", "PE_MatchBinary"};
    fprintf(to, "\n");
    if (!streamComp_.empty()) {
        fprintf(to, "StreamSizes:\n");
        for (uint32_t i = 0; i < streamComp_.size() && i < sizeof names / sizeof names[0]; ++i) fprintf(to, "%s %llu\n", names[i], (unsigned long long)streamComp_[i]);
        fprintf(to, "NDna: %llu\nNQua: %llu\nNReadIdToken: %llu\nNReadIdValue: %llu\n\n", (unsigned long long)(haveRaw_ ? rawComp_[0] : 0), (unsigned long long)(haveRaw_ ? rawComp_[1] : 0),
                (unsigned long long)(haveRaw_ ? rawComp_[2] : 0), (unsigned long long)(haveRaw_ ? rawComp_[3] : 0));
    }
    fprintf(to, "**** **** **** ****\n");
    fflush(to);
}

void ArchiveWriter::start(const std::string& prefix, const BinModuleConfigRaw& cfg)
{
    prefix_ = prefix;
    meta_ = fopen((prefix + ".cmeta").c_str(), "wb");
    data_ = fopen((prefix + ".cdata").c_str(), "w+b");      // (read access too: the blocks go in through a shared mapping)
    // (FileStreamWriter, fastore_bin/FileStream.cpp:305; ArchiveFileWriter opens the .cmeta first, fastore_pack/ArchiveFile.cpp:44-50)
    if (!meta_ || !data_) throw std::runtime_error("Cannot open file to write: " + prefix + (!meta_ ? ".cmeta" : ".cdata"));
    memset(&conf_, 0, sizeof conf_);                       // padding bytes are zero here (stack garbage in the reference)
    conf_.archType = cfg.archiveType; conf_.minParams = cfg.minimizer;
    conf_.quaParams.method = cfg.quaParams.method; conf_.quaParams.binaryThreshold = cfg.quaParams.binaryThreshold;
    conf_.quaParams.qvzOpts.verbose = cfg.quaParams.qvzOpts.verbose; conf_.quaParams.qvzOpts.stats = cfg.quaParams.qvzOpts.stats;
    conf_.quaParams.qvzOpts.uncompressed = cfg.quaParams.qvzOpts.uncompressed; conf_.quaParams.qvzOpts.distortion = cfg.quaParams.qvzOpts.distortion;
    conf_.quaParams.qvzOpts.D = cfg.quaParams.qvzOpts.D;    // the two char* members are process-local garbage: left null
    static const uint8_t zeros[24] = {0};
    if (fwrite(zeros, 1, 24, meta_) != 24) throw std::runtime_error("Cannot write " + prefix_ + ".cmeta");
    sizeStats_.start(cfg.archiveType, cfg.minimizer);
}

// Longest-processing-time-first over the bins' record totals, aware of what a rank's step is made of: a rank is done when its
// LONGEST stream is (one wavefront walks a PPMd stream from end to end: kLoneFactor records' worth of throughput work per
// record of its largest bin -- a lone stream runs at ~6 M symbols/s, the chip at ~5 G symbols/s when full) or when the SUM of
// its work is, whichever comes later.  Every bin goes where the job's makespan under that model ends up smallest; among
// equals, to the rank with the least sum.  (With the heaviest bins going first this differs from the plain rule only when a
// rank's sum outgrows its longest stream -- libraries far larger than 10 M reads -- but then it counts.)
std::vector<uint32_t> shardOwners(const std::vector<uint64_t>& weights, uint32_t world)
{
    const uint32_t n = (uint32_t)weights.size();
    std::vector<uint32_t> owner(n, 0), idx(n);
    if (world <= 1) return owner;
    for (uint32_t i = 0; i < n; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return weights[a] > weights[b]; });
    constexpr uint64_t kLoneFactor = 256;
    std::vector<uint64_t> load(world, 0), longest(world, 0);
    auto cost = [&](uint32_t r, uint64_t extra) { return std::max(kLoneFactor * std::max(longest[r], extra), load[r] + extra); };
    for (uint32_t i : idx) {
        const uint64_t w = weights[i] + 1;                         // (+1: empty weights still spread)
        uint32_t best = 0; uint64_t bestSpan = ~0ull;
        for (uint32_t r = 0; r < world; ++r) {
            uint64_t span = cost(r, w);
            for (uint32_t q = 0; q < world; ++q) if (q != r) span = std::max(span, cost(q, 0));
            if (span < bestSpan || (span == bestSpan && load[r] < load[best])) { best = r; bestSpan = span; }
        }
        owner[i] = best; load[best] += w; longest[best] = std::max(longest[best], w);
    }
    return owner;
}

void ArchiveWriter::startInMemory(const BinModuleConfigRaw& cfg)
{
    inMemory_ = true; prefix_ = "(held)";
    memset(&conf_, 0, sizeof conf_);
    conf_.archType = cfg.archiveType; conf_.minParams = cfg.minimizer;
    conf_.quaParams.method = cfg.quaParams.method; conf_.quaParams.binaryThreshold = cfg.quaParams.binaryThreshold;
    conf_.quaParams.qvzOpts.verbose = cfg.quaParams.qvzOpts.verbose; conf_.quaParams.qvzOpts.stats = cfg.quaParams.qvzOpts.stats;
    conf_.quaParams.qvzOpts.uncompressed = cfg.quaParams.qvzOpts.uncompressed; conf_.quaParams.qvzOpts.distortion = cfg.quaParams.qvzOpts.distortion;
    conf_.quaParams.qvzOpts.D = cfg.quaParams.qvzOpts.D;
    sizeStats_.start(cfg.archiveType, cfg.minimizer);
}

void ArchiveWriter::writeBlock(const uint8_t* data, uint64_t size, uint32_t signature)
{
    sizes_.push_back(size); sigs_.push_back(signature);
    sizeStats_.addBlock(data, size, signature);
    if (inMemory_) { held_.push_back(HeldBlock{signature, (uint64_t)mem_.size(), size}); mem_.insert(mem_.end(), data, data + size); dataBytes_ += size; return; }
    if (fwrite(data, 1, size, data_) != size) throw std::runtime_error("Cannot write " + prefix_ + ".cdata");
    dataBytes_ += size;
}

void ArchiveWriter::writeBlocks(const std::vector<const uint8_t*>& data, const std::vector<uint64_t>& sizes, const std::vector<uint32_t>& signatures, uint32_t threads)
{
    const size_t n = data.size();
    if (inMemory_ || n < 8 || threads < 2) { for (size_t i = 0; i < n; ++i) writeBlock(data[i], sizes[i], signatures[i]); return; }
    if (fflush(data_) != 0) throw std::runtime_error("Cannot write " + prefix_ + ".cdata");
    const off_t base = ftello(data_);
    if (base < 0) throw std::runtime_error("Cannot write " + prefix_ + ".cdata");
    std::vector<uint64_t> off(n + 1, 0);
    for (size_t i = 0; i < n; ++i) { off[i + 1] = off[i] + sizes[i]; sizes_.push_back(sizes[i]); sigs_.push_back(signatures[i]); sizeStats_.addBlock(data[i], sizes[i], signatures[i]); }
    const int fd = fileno(data_);
    // pieces of about equal bytes, one thread each
    const uint32_t t = (uint32_t)std::min<size_t>(threads, n);
    std::vector<size_t> cut(t + 1, n); cut[0] = 0;
    for (uint32_t k = 1; k < t; ++k) { const uint64_t want = off[n] / t * k; cut[k] = (size_t)(std::lower_bound(off.begin(), off.end(), want) - off.begin()); if (cut[k] > n) cut[k] = n; }
    // Through a shared mapping of the file's new extent when the file system allows it: write() calls on ONE file take
    // its lock in turn (~4 GB/s whatever the thread count), page-cache pages of a mapping are filled side by side.
    if (pre_ && (uint64_t)base < preLen_) {
        if (preTh_.joinable()) preTh_.join();                     // (long done: its pages were made while the device worked)
        // blocks that end inside the reserved extent go into its (already present) pages; what lies behind it -- the estimate
        // was short -- is written at its place
        uint8_t* dst = pre_ + base;
        parallelFor(t, t, [&](uint32_t k, uint32_t) {
            for (size_t i = cut[k]; i < cut[k + 1]; ++i) {
                if ((uint64_t)base + off[i + 1] <= preLen_) { memcpy(dst + off[i], data[i], sizes[i]); continue; }
                uint64_t done = 0;
                while (done < sizes[i]) {
                    const ssize_t w = pwrite(fd, data[i] + done, sizes[i] - done, base + (off_t)(off[i] + done));
                    if (w <= 0) throw std::runtime_error("Cannot write " + prefix_ + ".cdata");
                    done += (uint64_t)w;
                }
            }
        });
        if (fseeko(data_, base + (off_t)off[n], SEEK_SET) != 0) throw std::runtime_error("Cannot write " + prefix_ + ".cdata");
        dataBytes_ += off[n];
        return;
    }
    {
        const long pg = sysconf(_SC_PAGESIZE);
        const off_t mapFrom = base - (base % pg); const size_t mapLen = (size_t)(base - mapFrom) + off[n];
        // (the extent is reserved first: a full disk then ends in an error here, not in a bus error inside the mapping)
        if (off[n] > 0 && fallocate(fd, 0, base, (off_t)off[n]) == 0) {
            void* mp = mmap(nullptr, mapLen, PROT_READ | PROT_WRITE, MAP_SHARED, fd, mapFrom);
            if (mp != MAP_FAILED) {
                uint8_t* dst = (uint8_t*)mp + (base - mapFrom);
                parallelFor(t, t, [&](uint32_t k, uint32_t) { for (size_t i = cut[k]; i < cut[k + 1]; ++i) memcpy(dst + off[i], data[i], sizes[i]); });
                if (munmap(mp, mapLen) != 0) throw std::runtime_error("Cannot write " + prefix_ + ".cdata");
                if (fseeko(data_, base + (off_t)off[n], SEEK_SET) != 0) throw std::runtime_error("Cannot write " + prefix_ + ".cdata");
                dataBytes_ += off[n];
                return;
            }
        }
    }
    parallelFor(t, t, [&](uint32_t k, uint32_t) {
        for (size_t i = cut[k]; i < cut[k + 1]; ++i) {
            uint64_t done = 0;
            while (done < sizes[i]) {
                const ssize_t w = pwrite(fd, data[i] + done, sizes[i] - done, base + (off_t)(off[i] + done));
                if (w <= 0) throw std::runtime_error("Cannot write " + prefix_ + ".cdata");
                done += (uint64_t)w;
            }
        }
    });
    if (fseeko(data_, base + (off_t)off[n], SEEK_SET) != 0) throw std::runtime_error("Cannot write " + prefix_ + ".cdata");
    dataBytes_ += off[n];
}

// every write is checked: a full disk must end in "Error: Cannot write ..." (FSGPU_ERR_IO), not in a truncated archive
void ArchiveWriter::writeMeta(const std::string& prefix, const std::vector<uint64_t>& sizes, const std::vector<uint32_t>& sigs, const HeaderStats& head, const QvzModel& qvz)
{
    ArchiveWriter w;
    w.prefix_ = prefix; w.conf_ = conf_; w.sizes_ = sizes; w.sigs_ = sigs;
    w.meta_ = fopen((prefix + ".cmeta").c_str(), "wb");
    if (!w.meta_) throw std::runtime_error("Cannot open file to write: " + prefix + ".cmeta");
    static const uint8_t zeros[24] = {0};
    if (fwrite(zeros, 1, 24, w.meta_) != 24) throw std::runtime_error("Cannot write " + prefix + ".cmeta");
    w.finish(head, qvz);
}

void ArchiveWriter::finish(const HeaderStats& head, const QvzModel& qvz)
{
    if (inMemory_) return;                                 // held blocks: shardWrite places them
    const std::string what = "Cannot write " + prefix_ + ".cmeta";
    auto put = [&](const void* p, size_t n) { if (n && fwrite(p, 1, n, meta_) != n) throw std::runtime_error(what); };
    const uint64_t footerOffset = 24;
    const uint32_t count = (uint32_t)sizes_.size();
    put(&count, 4);
    put(sizes_.data(), 8 * sizes_.size());
    put(sigs_.data(), 4 * sigs_.size());
    put(&conf_, sizeof conf_);
    // quality data first, then the read-id field table (ArchiveFile.cpp:126-150)
    if (conf_.quaParams.method == MET_QVZ) {
        if (!qvz.present) throw std::runtime_error("QVZ archive without its codebook");
        put(qvz.footerBytes.data(), qvz.footerBytes.size());
    }
    if (conf_.archType.readsHaveHeaders) {
        std::vector<uint8_t> blob;
        serializeHeaderFields(head, conf_.archType.readType == READ_PE, blob);
        put(blob.data(), blob.size());
    }
    const off_t endPos = ftello(meta_);
    if (endPos < 0) throw std::runtime_error(what);
    const uint64_t footerSize = (uint64_t)endPos - footerOffset;
    if (fseeko(meta_, 0, SEEK_SET) != 0) throw std::runtime_error(what);
    put(&footerOffset, 8); put(&footerSize, 8);
    FILE* m = meta_; meta_ = nullptr;
    if (fclose(m) != 0) throw std::runtime_error(what);
    if (data_ && pre_) {
        // the reserved extent was an estimate: the archive ends where its last block ends
        const bool had = true; dropAhead();
        if (had && (fflush(data_) != 0 || ftello(data_) < 0 || ftruncate(fileno(data_), ftello(data_)) != 0)) throw std::runtime_error("Cannot write " + prefix_ + ".cdata");
    }
    FILE* d = data_; data_ = nullptr;
    if (d && fclose(d) != 0) throw std::runtime_error("Cannot write " + prefix_ + ".cdata");
}

// ------------------------------------------------------------------------------------------------
// Lanes: independent engine instances on the same GPU (own HIP stream, arena pool, buffers).  A batch is cut into
// slices; while the device codes slice k on one lane, the host threads run the front end of slice k+1, and the kernels
// of consecutive slices overlap on the device (the waves a draining kernel frees are taken by the next one).
fsengine::Device* Context::device()
{
    if (devAsync.load()) {
        std::lock_guard<std::mutex> g(devMx);
        if (devAsync.load()) { devError = devPending.get(); devAsync.store(false); }
    }
    if (!dev) throw std::runtime_error(devError.empty() ? std::string("no device") : devError);
    return dev;
}

fsengine::Device* Context::lane(uint32_t i)
{
    if (lanes.empty()) lanes.push_back(device());
    while (lanes.size() <= i) {
        fsengine::Device* d = nullptr; char e[256] = {0};
        if (fsengine::lane_create(device(), &d, e, sizeof e) != 0) {
            if (getenv("FS_TRACE")) fprintf(stderr, "[trace] no further engine lane: %s\n", e);
            return nullptr;
        }
        lanes.push_back(d);
    }
    return lanes[i];
}

// The read-id field table of a library as fs_tokenise_ids reads it (device_types.h: IdField); also the numbers of
// (symbol, context) pairs a read id gives at most: token pairs, value pairs
static void idFieldBlob(const HeaderStats& head, std::vector<uint8_t>& blob, uint32_t& tokPerRead, uint32_t& valPerRead)
{
    const uint32_t nf = (uint32_t)head.fields.size();
    tokPerRead = valPerRead = 0;
    std::vector<fsdev::IdField> F(nf);
    size_t at = 8 + nf * sizeof(fsdev::IdField);
    std::vector<uint8_t> lists;                                   // value lists, then the value bytes
    for (uint32_t i = 0; i < nf; ++i) {
        const HeaderField& f = head.fields[i];
        fsdev::IdField& d = F[i]; memset(&d, 0, sizeof d);
        d.separator = (uint8_t)f.separator; d.is_const = f.isConst ? 1 : 0; d.is_numeric = f.isNumeric ? 1 : 0; d.min_value = f.minValue;
        if (f.isConst) continue;
        if (f.isNumeric) {
            const int32_t valueRange = (int32_t)(f.maxValue - f.minValue);      // as IHeaderStoreBase::CompressReadId computes it
            d.plog = (uint8_t)intLog((uint64_t)(int64_t)valueRange, 256);
            valPerRead += d.plog + 1u;
        } else { d.n_values = (uint32_t)f.possibleValues.size(); ++tokPerRead; }
    }
    for (uint32_t i = 0; i < nf; ++i) {
        const HeaderField& f = head.fields[i];
        if (f.isConst || f.isNumeric) continue;
        F[i].values_off = (uint32_t)(at + lists.size());
        const size_t listAt = lists.size(); lists.resize(listAt + 8 * f.possibleValues.size());
        for (size_t v = 0; v < f.possibleValues.size(); ++v) {
            const uint32_t off = (uint32_t)(at + lists.size()), len = (uint32_t)f.possibleValues[v].size();
            memcpy(lists.data() + listAt + 8 * v, &off, 4); memcpy(lists.data() + listAt + 8 * v + 4, &len, 4);
            lists.insert(lists.end(), f.possibleValues[v].begin(), f.possibleValues[v].end());
        }
        while (lists.size() & 7u) lists.push_back(0);
    }
    blob.assign(at + lists.size() + 8, 0);
    memcpy(blob.data(), &nf, 4);
    if (nf) memcpy(blob.data() + 8, F.data(), nf * sizeof(fsdev::IdField));
    if (!lists.empty()) memcpy(blob.data() + at, lists.data(), lists.size());
}

// The window searches of host thread `tid`: its own matcher lane (stream + buffers), created on the first bin
MatchFn Context::matcherFor(uint32_t tid)
{
    if (!deviceMatcher) return MatchFn();
    if (matchLanes.size() <= tid) throw std::runtime_error("matcher lanes not sized");       // (sized by the callers before their threads start)
    return [this, tid](const uint8_t* seq, size_t seqBytes, const fsdev::PackedDna* packed, const fsdev::MatchRead* reads, size_t nReads, const fsdev::MatchCall* calls, size_t nCalls,
                       const uint32_t* warm, size_t nWarm, const fsdev::MatchParams& mp, fsdev::MatchRow* rows) -> bool {
        // (a device still on its way -- one-shot contexts -- is waited for a little: the heaviest bins, searched first, gain
        // most from it; should it take longer -- a driver clearing memory -- the host scan gives the same rows)
        if (!deviceReadyWithin(300)) return false;
        fsengine::Device* dev = device();
        if (!matchLanes[tid]) {
            // a thread's first search: its lane, with room for the batch's largest bin at once (from nothing: no buffer
            // is freed here, so no wait for running kernels) -- the host threads do this side by side, in their first bins
            if (fsengine::match_lane_create(dev, &matchLanes[tid]) != 0) throw std::runtime_error(std::string("device: ") + dev->err);
            if (matchReserve.reads && fsengine::match_lane_reserve(dev, matchLanes[tid], matchReserve.reads, matchReserve.seqBytes, matchReserve.calls, matchReserve.warm) != 0)
                throw std::runtime_error(std::string("device: ") + dev->err);
        }
        const double t0 = nowMs(); double kms = 0;
        struct InFlight { std::atomic<int>& n; explicit InFlight(std::atomic<int>& x) : n(x) { ++n; } ~InFlight() { --n; } } inFlight(searchesInFlight);
        if (fsengine::match_reads(dev, matchLanes[tid], seq, seqBytes, packed, reads, nReads, calls, nCalls, warm, nWarm, mp, rows, &kms) != 0) throw std::runtime_error(std::string("device: ") + dev->err);
        matchBasesUp += packed ? packed->bytes + nReads * sizeof(fsdev::PackedRead) : seqBytes; if (packed) matchUnpackedReads += nReads;
        matchedReads += nReads; matchUs += (uint64_t)((nowMs() - t0) * 1e3); matchKernelUs += (uint64_t)(kms * 1e3);
        return true;
    };
}

MateFn Context::mateMatcherFor(uint32_t tid)
{
    if (!deviceMatcher) return MateFn();
    if (matchLanes.size() <= tid) throw std::runtime_error("matcher lanes not sized");
    return [this, tid](const uint8_t* seq, size_t seqBytes, const fsdev::MatePair* pairs, size_t nPairs, const uint32_t* validBits, size_t validWords,
                       const fsdev::MateParams& mp, fsdev::MateRow* rows) -> bool {
        if (!deviceReadyWithin(300)) return false;
        fsengine::Device* dev = device();
        if (!matchLanes[tid]) { if (fsengine::match_lane_create(dev, &matchLanes[tid]) != 0) throw std::runtime_error(std::string("device: ") + dev->err); }
        const double t0 = nowMs(); double kms = 0;
        if (fsengine::match_mates(dev, matchLanes[tid], seq, seqBytes, pairs, nPairs, validBits, validWords, mp, rows, &kms) != 0) throw std::runtime_error(std::string("device: ") + dev->err);
        matedPairs += nPairs; mateUs += (uint64_t)((nowMs() - t0) * 1e3); mateKernelUs += (uint64_t)(kms * 1e3);
        return true;
    };
}

// The device's mate searches without the host threads waiting for them (FS_DEVICE_MATES=2): a bin's pairs are handed over when
// its walk is done (fs::PendingPairs), two worker threads gather what has come in, search a whole batch of bins in one launch
// (fs_match_mates: a workgroup per bin), write the bins' mate streams from the rows and report the bins complete.
struct Context::MateDispatcher {
    Context& c; std::mutex mx; std::condition_variable cv; std::vector<std::unique_ptr<PendingPairs>> q; bool closing = false; uint32_t inFlight = 0;
    std::vector<std::thread> workers; std::vector<fsengine::MatchLane*> lanes;
    std::atomic<uint32_t> launches{0}, binsSent{0}, maxBins{0}; std::atomic<uint64_t> pairsSent{0}, kernelUs{0}, maxLaunchUs{0};       // (FS_TRACE)
    explicit MateDispatcher(Context& ctx) : c(ctx)
    {
        lanes.assign(2, nullptr);
        for (uint32_t w = 0; w < 2; ++w) workers.emplace_back([this, w]() { run(w); });
    }
    ~MateDispatcher()
    {
        { std::lock_guard<std::mutex> g(mx); closing = true; }
        cv.notify_all();
        for (auto& t : workers) t.join();
        for (auto* l : lanes) if (l) fsengine::match_lane_destroy(l);
        if (getenv("FS_TRACE")) fprintf(stderr, "[trace] mate searches: %u launches, %u bins, %llu pairs, %.1f ms of kernels, longest launch %.1f ms, most bins in a launch %u\n", launches.load(), binsSent.load(), (unsigned long long)pairsSent.load(), kernelUs.load() / 1e3, maxLaunchUs.load() / 1e3, maxBins.load());
    }
    bool push(std::unique_ptr<PendingPairs> pp)
    {
        { std::lock_guard<std::mutex> g(mx); if (closing) return false; q.push_back(std::move(pp)); }
        cv.notify_one();
        return true;
    }
    static bool sameParams(const fsdev::MateParams& a, const fsdev::MateParams& b) { return memcmp(&a, &b, sizeof a) == 0; }
    void run(uint32_t w)
    {
        for (;;) {
            std::vector<std::unique_ptr<PendingPairs>> batch;
            {
                std::unique_lock<std::mutex> lk(mx);
                cv.wait(lk, [&]() { return !q.empty() || closing; });
                if (q.empty()) return;                       // closing, nothing left
                // (a moment for more bins to arrive: a launch carries what is there, up to 64 bins of one archive's parameters)
                if (q.size() < 8 && !closing) { lk.unlock(); std::this_thread::sleep_for(std::chrono::microseconds(300)); lk.lock(); }
                uint64_t bytes = 0;
                for (size_t i = 0; i < q.size() && batch.size() < 64;) {
                    if (!batch.empty() && (!sameParams(q[i]->mp, batch[0]->mp) || q[i]->validBits != batch[0]->validBits || bytes + q[i]->seqBytes > (1ull << 30))) { ++i; continue; }
                    bytes += q[i]->seqBytes; batch.push_back(std::move(q[i])); q.erase(q.begin() + (ptrdiff_t)i);
                }
                if (batch.empty()) continue;                 // (the other worker took them during that moment)
                inFlight += (uint32_t)batch.size();
            }
            std::string err;
            try {
                fsengine::Device* dev = c.device();
                if (!lanes[w] && fsengine::match_lane_create(dev, &lanes[w], true) != 0) throw std::runtime_error(std::string("device: ") + dev->err);
                std::vector<fsengine::MateBatchJob> jobs(batch.size());
                uint64_t nPairs = 0;
                for (size_t j = 0; j < batch.size(); ++j) {
                    PendingPairs& pp = *batch[j];
                    pp.rows.resize(pp.pairs.size());
                    jobs[j] = fsengine::MateBatchJob{pp.seq, pp.seqBytes, pp.pairs.data(), pp.pairs.size(), pp.rows.data()};
                    nPairs += pp.pairs.size();
                }
                const double t0 = nowMs(); double kms = 0;
                if (fsengine::match_mates_batch(dev, lanes[w], jobs.data(), jobs.size(), batch[0]->validBits.data(), batch[0]->validBits.size(), batch[0]->mp, &kms) != 0) throw std::runtime_error(std::string("device: ") + dev->err);
                c.matedPairs += nPairs; c.mateUs += (uint64_t)((nowMs() - t0) * 1e3); c.mateKernelUs += (uint64_t)(kms * 1e3);
                ++launches; binsSent += (uint32_t)batch.size(); pairsSent += nPairs; kernelUs += (uint64_t)(kms * 1e3);
                { uint64_t m = maxLaunchUs.load(); while ((uint64_t)(kms * 1e3) > m && !maxLaunchUs.compare_exchange_weak(m, (uint64_t)(kms * 1e3))) {} }
                { uint32_t m = maxBins.load(); while ((uint32_t)batch.size() > m && !maxBins.compare_exchange_weak(m, (uint32_t)batch.size())) {} }
                for (auto& pp : batch) emitPairsFromRows(*pp);
            } catch (const std::exception& e) { err = e.what(); }
            for (auto& pp : batch) pp->done(err.empty() ? nullptr : err.c_str());
            { std::lock_guard<std::mutex> g(mx); inFlight -= (uint32_t)batch.size(); }
            cv.notify_all();
        }
    }
};

void Context::mateMatcherCheck(const std::string& inPrefix, uint64_t& pairs, uint64_t& differing)
{
    pairs = differing = 0;
    (void)device();
    BinFile bf; bf.open(inPrefix, par.minBinSize);
    ArchiveParams arch; arch.cfg = bf.config(); arch.head = bf.headData(); arch.qvz = bf.qvz();
    if (arch.cfg.archiveType.readType != READ_PE) return;
    const std::vector<uint32_t>& sigs = bf.stdSignatures();
    std::mutex mx;
    const bool keep = deviceMatcher; deviceMatcher = true;
    if (matchLanes.size() < hostThreads) matchLanes.resize(hostThreads, nullptr);
    parallelFor((uint32_t)sigs.size(), std::min<uint32_t>(hostThreads, 8u), [&](uint32_t k, uint32_t tid) {
        Batch b; bf.unpack(sigs[k], b, true);
        BinEncoder enc(par);
        uint64_t r = 0, d = 0;
        enc.checkMateMatcher(b, b, b.bins.at(0), arch, mateMatcherFor(tid), r, d);
        std::lock_guard<std::mutex> g(mx); pairs += r; differing += d;
    });
    deviceMatcher = keep;
}

// Parity check of the device matcher on every standard bin of a library: host scan vs device, row by row
void Context::matcherCheck(const std::string& inPrefix, uint64_t& reads, uint64_t& differing)
{
    reads = differing = 0;
    (void)device();                                                  // (a device still on its way is waited for: this call is about it)
    BinFile bf; bf.open(inPrefix, par.minBinSize);
    ArchiveParams arch; arch.cfg = bf.config(); arch.head = bf.headData(); arch.qvz = bf.qvz();
    const std::vector<uint32_t>& sigs = bf.stdSignatures();
    std::mutex mx;
    const bool keep = deviceMatcher; deviceMatcher = true;
    if (matchLanes.size() < hostThreads) matchLanes.resize(hostThreads, nullptr);
    parallelFor((uint32_t)sigs.size(), std::min<uint32_t>(hostThreads, 8u), [&](uint32_t k, uint32_t tid) {
        Batch b; bf.unpack(sigs[k], b, true, deviceUnpack());
        BinEncoder enc(par);
        uint64_t r = 0, d = 0;
        enc.checkMatcher(b, b, b.bins.at(0), arch, matcherFor(tid), r, d);
        std::lock_guard<std::mutex> g(mx); reads += r; differing += d;
    });
    deviceMatcher = keep;
}

// Parity harness of the emission kernels (fs_emit_count / _scan / _write, fs_rle_binary, fs_rle0): every standard bin of a library
// through the walk twice -- once writing the streams that hold bases itself (the host restatement of CompressHardRead ...
// StoreContigDefinition and of the run-length coders), once leaving ops -- and the ops through the device; the eleven streams
// (seven in single-end bins) byte for byte.
void Context::emitCheck(const std::string& inPrefix, uint64_t& ops, uint64_t& streamsCompared, uint64_t& differing)
{
    ops = streamsCompared = differing = 0;
    (void)device();
    BinFile bf; bf.open(inPrefix, par.minBinSize);
    ArchiveParams arch; arch.cfg = bf.config(); arch.head = bf.headData(); arch.qvz = bf.qvz();
    static const uint32_t streamOf[fsdev::ECH_COUNT + 1] = {S_HardReads, S_LettersX, S_Match, S_MatchBinary, S_CMatch, S_CLetters, S_HardPE, S_LettersXPE, S_MatchRlePE, S_MatchBinaryPE, S_LzId};
    for (uint32_t sig : bf.stdSignatures()) {
        Batch b; bf.unpack(sig, b, true);
        BinEncoder enc(par);
        BinStreams host, dev;
        enc.setDeviceEmit(false); enc.encodeLz(b, b, b.bins.at(0), arch, host);
        enc.setDeviceEmit(true); enc.encodeLz(b, b, b.bins.at(0), arch, dev);
        if (!dev.deviceEmit) throw std::runtime_error("the walk left no ops");
        const bool pe = dev.nStreams == S_PE_COUNT;
        // one job: the bin's bases, contig bytes, ops and ids; every stream's room by the walk's bounds
        fsdev::EmitJob job; memset(&job, 0, sizeof job);
        uint64_t off = 0, outBytes = 0;
        job.seq_off = off; job.seq_bytes = (uint32_t)(dev.emitSeqHi - dev.emitSeqLo); off += (job.seq_bytes + 15u) & ~15ull;
        job.contig_off = off; job.contig_bytes = (uint32_t)dev.contigBytes.size(); off += (job.contig_bytes + 15u) & ~15ull;
        fsdev::EmitPlan plan; plan.n_jobs = 1; plan.n_ops = (uint32_t)dev.emitOps.size(); plan.n_ids = (uint32_t)dev.lzIds.size();
        plan.jobs_off = off; off += (sizeof job + 15u) & ~15ull;
        plan.ops_off = off; off += (dev.emitOps.size() * sizeof(fsdev::EmitOp) + 15u) & ~15ull;
        plan.ids_off = off; off += (4ull * dev.lzIds.size() + 15u) & ~15ull;
        job.n_ops = plan.n_ops; job.n_ids = plan.n_ids;
        job.sig_len = arch.cfg.minimizer.signatureLen; job.begin_cut = par.beginCut; job.end_cut = par.endCut;
        memset(job.dna_to_idx, 0xFF, sizeof job.dna_to_idx);
        for (int q = 0; q < 5; ++q) job.dna_to_idx[(uint8_t)arch.cfg.minimizer.dnaSymbolOrder[q] & 127u] = (uint8_t)q;
        std::vector<fsdev::StreamItem> items;
        for (uint32_t c = 0; c <= fsdev::ECH_COUNT; ++c) {
            const bool peChannel = c >= fsdev::ECH_HARD_PE && c < fsdev::ECH_COUNT;
            if (peChannel && !pe) { job.item[c] = 0xFFFFFFFFu; continue; }
            const bool bits = c == fsdev::ECH_MATCH_BITS || c == fsdev::ECH_CMATCH_BITS || c == fsdev::ECH_MATCH_BITS_PE;
            const uint64_t unit = (c == fsdev::ECH_COUNT || bits || c == fsdev::ECH_HARD || c == fsdev::ECH_HARD_PE) ? 1u : 2u;
            const uint64_t bound = dev.emitBound[c] + 2u;
            fsdev::StreamItem it; memset(&it, 0, sizeof it); it.in_len = (uint32_t)bound;
            job.item[c] = (uint32_t)items.size(); job.cap[c] = (uint32_t)bound; job.out_off[c] = outBytes; outBytes += (unit * bound + 2u + 15u) & ~15ull;
            if (bits) { job.raw_off[c] = outBytes; outBytes += (bound + 15u) & ~15ull; }
            items.push_back(it);
        }
        plan.out_bytes = outBytes;
        std::vector<uint8_t> input(off + 16, 0);
        memcpy(input.data() + job.seq_off, dev.emitSeq, job.seq_bytes);
        if (job.contig_bytes) memcpy(input.data() + job.contig_off, dev.contigBytes.data(), job.contig_bytes);
        memcpy(input.data() + plan.jobs_off, &job, sizeof job);
        if (plan.n_ops) memcpy(input.data() + plan.ops_off, dev.emitOps.data(), dev.emitOps.size() * sizeof(fsdev::EmitOp));      // (pad2[0] = job 0 already)
        if (plan.n_ids) memcpy(input.data() + plan.ids_off, dev.lzIds.data(), 4ull * dev.lzIds.size());
        std::vector<std::vector<std::vector<uint8_t>>> got;
        if (fsengine::emit_streams_raw(device(), input.data(), off, plan, items, got) != 0) throw std::runtime_error(std::string("device: ") + device()->err);
        ops += plan.n_ops;
        for (uint32_t c = 0; c <= fsdev::ECH_COUNT; ++c) {
            if (job.item[c] == 0xFFFFFFFFu) continue;
            ++streamsCompared;
            if (got[0][c] != host.s[streamOf[c]]) ++differing;
        }
        // (every other stream is the walk's own either way)
        for (uint32_t s = 0; s < dev.nStreams; ++s) {
            bool devStream = false;
            for (uint32_t c = 0; c <= fsdev::ECH_COUNT; ++c) devStream = devStream || streamOf[c] == s;
            if (!devStream && dev.s[s] != host.s[s]) ++differing;
            if (devStream && !dev.s[s].empty()) ++differing;
        }
    }
}

void Context::tokeniserCheck(const std::string& inPrefix, uint64_t& ids, uint64_t& differingBins)
{
    ids = differingBins = 0;
    BinFile bf; bf.open(inPrefix, par.minBinSize);
    if (!bf.usesHeaders()) return;
    const HeaderStats head = bf.headData();
    std::vector<uint8_t> table; uint32_t tokPer = 0, valPer = 0;
    idFieldBlob(head, table, tokPer, valPer);
    for (uint32_t sig : bf.stdSignatures()) {
        const BinInfo& bi = bf.bins().at(sig);
        // host: unpacked headers in stored order through IHeaderStoreBase::CompressReadId's restatement
        Batch hb; bf.unpack(sig, hb, true);
        std::vector<uint8_t> tokH, valH;
        for (const Rec& r : hb.recs) compressReadId(head, hb.head.data() + r.headOff, r.headLen, tokH, valH);
        // device: the same records from the packed headers
        Batch pb, pg;
        pb.seq.resize(bi.totalRawDnaSize); pb.qua.resize(bi.totalRawDnaSize); pb.recs.resize(bi.totalRecordsCount);
        pb.headPacked.resize(bi.totalHeadSize + 16); pb.headBit.resize(bi.totalRecordsCount);
        bf.unpackPlaced(sig, pb, 0, 0, 0, pg, -1, 0);
        const size_t n = pb.recs.size();
        uint64_t off = 0;
        const uint64_t tabOff = off; off += (table.size() + 15) & ~15ull;
        const uint64_t headOff = off; off += (bi.totalHeadSize + 8 + 15) & ~15ull;
        const uint64_t jobsOff = off; off += (sizeof(fsdev::IdJob) + 15) & ~15ull;
        const uint64_t strOff = off; off += (n * sizeof(fsdev::IdString) + 15) & ~15ull;
        std::vector<uint8_t> input(off + 16, 0);
        memcpy(input.data() + tabOff, table.data(), table.size());
        memcpy(input.data() + headOff, pb.headPacked.data(), bi.totalHeadSize);
        fsdev::IdJob job; memset(&job, 0, sizeof job);
        job.first = 0; job.count = (uint32_t)n; job.tok_item = 0; job.val_item = 1; job.table_off = tabOff;
        job.tok_out = 0; job.val_out = (2ull * n * tokPer + 31u) & ~15ull;
        memcpy(input.data() + jobsOff, &job, sizeof job);
        fsdev::IdString* ss = (fsdev::IdString*)(input.data() + strOff);
        for (size_t i = 0; i < n; ++i) { ss[i].src_bit = 8ull * headOff + pb.headBit[i]; ss[i].len = pb.recs[i].headLen; ss[i].pad = 0; }
        fsdev::IdPlan plan; plan.jobs_off = jobsOff; plan.strings_off = strOff; plan.n_jobs = 1; plan.n_strings = (uint32_t)n;
        plan.out_bytes = job.val_out + ((2ull * n * valPer + 31u) & ~15ull);
        std::vector<std::vector<uint8_t>> tok, val;
        if (fsengine::tokenise_ids_raw(device(), input.data(), off, plan, tok, val) != 0) throw std::runtime_error(std::string("device: ") + device()->err);
        ids += n;
        if (tok[0] != tokH || val[0] != valH) ++differingBins;
    }
}

void Context::equalizeNow()
{
    if (equalizeLanes < 2) return;
    const uint32_t n = std::min<uint32_t>(equalizeLanes, (uint32_t)lanes.size());
    for (uint32_t l = 0; l < n; ++l) (void)fsengine::staging_buffer(lanes[l], equalizeStage);
    if (fsengine::lanes_equalize(lanes.data(), n) != 0) throw std::runtime_error(std::string("device: ") + lanes[0]->err);
    equalizeLanes = 0;
}

void Context::gatherBlocks()
{
    const uint32_t nBins = (uint32_t)blockSizes.size();
    uint64_t total = 0;
    for (uint64_t v : blockSizes) total += v;
    blocks.resize(total);
    uint64_t off = 0;
    for (uint32_t b = 0; b < nBins; ++b) { memcpy(blocks.data() + off, blockData(b), blockSizes[b]); off += blockSizes[b]; }
}

void Context::compressBatch(const Batch& batch, const std::vector<uint32_t>& binArch)
{
    const uint32_t nBins = (uint32_t)batch.bins.size();
    std::vector<uint64_t> weight(nBins);
    for (uint32_t b = 0; b < nBins; ++b) weight[b] = batch.bins[b].recCount;
    binBases.assign(nBins, 0);
    for (uint32_t b = 0; b < nBins; ++b) binBases[b] = batch.bins[b].rawDnaSize;
    compressBins(nBins, weight, binArch, [&](uint32_t b, BinEncoder& enc, BinStreams& out, BinIn& info, uint64_t& recBytes) {
        info = batch.bins[b];
        enc.encodeLz(batch, info, archives[binArch[b]], out);
        uint64_t a = 0;
        for (uint32_t r = info.recBegin; r < info.recBegin + info.recCount; ++r) { const Rec& rc = batch.recs[r]; a += 2ull * (rc.seqLen + rc.auxLen) + rc.headLen; }
        recBytes = a;
    });
}

// The front end of the bins runs as one stream of tasks over the host threads (largest bins first); the bins are cut
// into slices, and the moment the last bin of a slice is done its lane thread builds the slice's stream items, stages
// them and runs the device call -- so the host front end of the later slices, the staging and the device work of the
// earlier ones all overlap, and the kernels of consecutive slices overlap on the device (shared arena pool).
enum : uint32_t { kMaxLanes = 14 };      // one HIP stream each, on its own hardware queue (GPU_MAX_HW_QUEUES=16; measured: 24 queues make launches wait for each other again)

void Context::compressBins(uint32_t nBins, const std::vector<uint64_t>& weight, const std::vector<uint32_t>& binArch, const BinProducer& produce)
{
    using namespace fsdev;
    blocks.clear(); blockSizes.assign(nBins, 0); blockSlice.assign(nBins, 0); blockOff.assign(nBins, 0);
    if (nBins == 0) return;
    const double t0 = nowMs();
    const bool trace = getenv("FS_TRACE") != nullptr;
    if (streamPool.size() < nBins) streamPool.resize(nBins);
    std::vector<BinStreams>& st = streamPool;
    std::vector<BinIn> info(nBins); std::vector<uint64_t> recBytes(nBins, 0);
    // largest bins first: their streams are the longest (a launch ends with its longest stream) and the front end of
    // a bin is sequential, so a big bin started last would be the tail on the host too
    std::vector<uint32_t> byWork(nBins);
    for (uint32_t b = 0; b < nBins; ++b) byWork[b] = b;
    std::sort(byWork.begin(), byWork.end(), [&](uint32_t x, uint32_t y) { return weight[x] > weight[y]; });
    uint64_t totalW = 0;
    for (uint64_t w : weight) totalW += w;
    // Slices.  The device step ends with the longest stream of the batch (a PPMd stream is serial: one wavefront walks
    // it from end to end), so the biggest bins must reach the device as early as possible: the first slice is what the
    // host threads finish in their first round -- one bin each --, the rest of the weight is cut into equal parts that
    // become ready one after the other while the earlier ones are coded.  Up to eight launches are in flight (one per
    // lane); they take their arenas from the pool's slot maps.
    std::vector<uint32_t> cut{0};
    const bool autoSlices = cfg.pipeline_slices == 0;
    // (one bin per CORE in the first round -- the threads beyond the cores joining as bins finish -- was tried: the first slice
    // goes up 25 ms earlier, the second one, whose streams are nearly as long, later: no gain, profiles/r02_yy_first_round.txt)
    const uint32_t firstRound = hostThreads;
    const uint32_t sliceRound = sliceThreads ? std::min(sliceThreads, hostThreads) : hostThreads;      // (the bins of a first slice)
    const uint32_t wantSlices = cfg.pipeline_slices ? std::min(cfg.pipeline_slices, nBins) : ((nBins >= 64 && totalW >= 200000) ? 14u : 1u);
    if (wantSlices > 1) {
        std::vector<double> share;
        if (!autoSlices) {
            uint64_t firstW = 0; const uint32_t firstBins = std::min<uint32_t>(std::max(1u, sliceRound), nBins / wantSlices);
            for (uint32_t i = 0; i < firstBins; ++i) firstW += weight[byWork[i]];
            const double f = std::min(0.5, (double)firstW / (double)std::max<uint64_t>(1, totalW));
            share.assign(wantSlices, (1.0 - f) / (wantSlices - 1)); share[0] = f;
        }
        if (!share.empty()) {
            double shareSum = 0; for (double v : share) shareSum += v;
            uint64_t acc = 0; uint32_t k = 0; double upTo = share[0];
            for (uint32_t i = 0; i < nBins && k + 1 < wantSlices; ++i) {
                acc += weight[byWork[i]];
                if ((double)acc >= upTo / shareSum * (double)totalW && i + 1 < nBins) { cut.push_back(i + 1); ++k; upTo += share[k]; }
            }
        } else {
            // Default rule.  A slice is ready when the host has been through its last bin, and then needs the device for as
            // long as its longest stream takes; the host works through the bins at a steady rate, heaviest first.  So the
            // first slice is exactly the bins the host threads take in their first round (one each: ready when the
            // heaviest bin is), the middle slices grow, and the last ones -- small bins only, short streams -- shrink
            // again, so that little device work is left when the host is done.
            const uint32_t firstBins = std::min<uint32_t>(std::max(1u, sliceRound), std::max(1u, nBins / wantSlices));
            uint64_t firstW = 0;
            for (uint32_t i = 0; i < firstBins; ++i) firstW += weight[byWork[i]];
            // (measured on the BASELINE library, profiles/r02_ae_slice_weights.txt: the first round goes up as TWO slices -- the
            // heavier half does not wait for the other's last bin --, and the slices right behind it stay small: the second-round
            // bins reach the device as they finish instead of waiting for 20 others, whose streams are nearly as long as the
            // first round's.  With one 24-bin first slice and 21 / 24 / 34 behind it the step ended with the SECOND slice.)
            static const double upToFrac[] = {0.0, 0.0, 0.05, 0.10, 0.167, 0.257, 0.369, 0.504, 0.639, 0.762, 0.863, 0.942, 0.981, 1.0};   // of the weight behind the first round
            const uint32_t half = firstBins >= 2u && wantSlices >= 4u ? (firstBins + 1u) / 2u : 0u;
            if (half) cut.push_back(half);
            cut.push_back(firstBins);
            uint64_t acc = 0; uint32_t k = half ? 2u : 1u;
            const double rest = (double)(totalW - firstW);
            for (uint32_t i = firstBins; i < nBins && k + 1 < wantSlices; ++i) {
                acc += weight[byWork[i]];
                if ((double)acc >= upToFrac[k] * rest && i + 1 < nBins && i + 1 > cut.back()) { cut.push_back(i + 1); ++k; }
            }
        }
    }
    cut.push_back(nBins);
    const uint32_t nSlices = (uint32_t)cut.size() - 1;
    // never more than kMaxLanes kernels in flight unless the caller insists (pipeline_lanes); extra slices queue behind
    const uint32_t wantLanes = nSlices > 1 ? std::min<uint32_t>(nSlices, cfg.pipeline_lanes ? cfg.pipeline_lanes : kMaxLanes) : 1;
    // lanes of an earlier batch are there; a fresh context (the CLI) makes lane 0 here and the others beside the front end
    // (a stream, events: 10-20 ms each), handing each to the slices as it comes
    const bool devPendingNow = !deviceReadyWithin(0);               // (one-shot context whose device is still being made: the lane maker waits for it)
    if (!devPendingNow) (void)lane(0);
    lanes.reserve(std::max<size_t>(lanes.capacity(), 64));          // (slice threads read lanes[] while the maker appends)
    const uint32_t haveLanes = std::min<uint32_t>(wantLanes, (uint32_t)lanes.size());
    uint32_t nLanes = haveLanes;

    // Lanes of the previous batch get the buffers of its best-equipped one now (not at its end: a one-shot run -- the CLI --
    // would pay for pinned and device memory it never uses): whichever slice a lane gets, nothing has to grow -- and so to
    // be freed, which waits for every kernel in flight -- while the long streams are being coded.
    if (!devPendingNow) equalizeNow();
    struct Slice {
        std::vector<StreamItem> items; std::vector<BlockPlan> plans; std::vector<uint64_t> sizes;
        fsengine::BatchTiming timing; std::string err; std::thread th; double tReady = 0, tSubmit = 0, tDone = 0, bufMs = 0;
        std::mutex mx; std::condition_variable cv; uint32_t pending = 0; bool done = false; int lane = -1; uint64_t inBytes = 0;
    };
    // a slice takes whichever lane is free when its bins are ready (the early slices hold theirs for the longest streams)
    std::mutex laneMx; std::condition_variable laneCv; std::vector<uint32_t> freeLanes;
    // (a fresh context: the maker also gives every lane its pinned staging buffer, sized for the slice that will most likely
    // take it -- slice i becomes ready i-th and takes the i-th lane --, before a slice can have it: 40-70 ms of hipHostMalloc
    // per lane that would otherwise sit between a slice's front end and its upload)
    std::vector<uint64_t> sliceEst(nSlices, 0);
    if (stageEstimate.size() == nBins) for (uint32_t si = 0; si < nSlices; ++si) for (uint32_t k = cut[si]; k < cut[si + 1]; ++k) sliceEst[si] += stageEstimate[byWork[k]];
    stageEstimate.clear();                                          // (it describes this call's bins only)
    const bool makeLanes = haveLanes < wantLanes;
    const bool fresh0 = makeLanes && (haveLanes == 0 || (haveLanes == 1 && lanes[0]->hStage == nullptr && sliceEst[0] != 0));
    for (uint32_t l = nLanes; l-- > 0;) if (!(fresh0 && l == 0)) freeLanes.push_back(l);
    std::vector<Slice> slices(nSlices);
    std::vector<uint32_t> sliceOf(nBins);
    for (uint32_t si = 0; si < nSlices; ++si) { slices[si].pending = cut[si + 1] - cut[si]; for (uint32_t k = cut[si]; k < cut[si + 1]; ++k) sliceOf[k] = si; }
    sliceBlocks.resize(std::max<size_t>(sliceBlocks.size(), nSlices));
    std::atomic<bool> abort(false);
    std::thread laneMaker;
    std::atomic<uint32_t> madeLanes(haveLanes);
    std::string makerErr;
    if (makeLanes) laneMaker = std::thread([&]() {
        auto prep = [&](fsengine::Device* d, uint32_t l) { if (l < nSlices && sliceEst[l]) (void)fsengine::staging_buffer(d, sliceEst[l] - sliceEst[l] / 8 /* the estimate runs ~8 % high and the buffer adds a quarter */); };
        fsengine::Device* first = nullptr;
        try {
            first = device();                                       // waits for a device that is still being made
            if (haveLanes == 0) { std::lock_guard<std::mutex> lk(laneMx); (void)lane(0); }
        } catch (const std::exception& e) {
            makerErr = e.what(); abort.store(true);
            for (Slice& sl : slices) { std::lock_guard<std::mutex> lk(sl.mx); sl.cv.notify_all(); }
            laneCv.notify_all();
            return;
        }
        if (fresh0) {
            prep(lanes[0], 0);
            { std::lock_guard<std::mutex> lk(laneMx); freeLanes.push_back(0); }
            laneCv.notify_one(); madeLanes = std::max<uint32_t>(madeLanes.load(), 1u);
        }
        for (uint32_t l = std::max<uint32_t>(haveLanes, 1u); l < wantLanes; ++l) {
            fsengine::Device* d = nullptr; char e[256] = {0};
            if (fsengine::lane_create(first, &d, e, sizeof e) != 0) { if (trace) fprintf(stderr, "[trace] no further engine lane: %s\n", e); break; }
            prep(d, l);
            { std::lock_guard<std::mutex> lk(laneMx); lanes.push_back(d); freeLanes.insert(freeLanes.begin(), l); }
            laneCv.notify_one(); madeLanes = l + 1;
        }
    });

    std::atomic<uint32_t> slicesWithLane(0), slicesStaged(0);
    std::thread stagedReleaser; std::mutex releaserMx;
    auto runSlice = [&](uint32_t si) {
        Slice& S = slices[si];
        {   // the slice's bins have all been through the front end
            std::unique_lock<std::mutex> lk(S.mx);
            S.cv.wait(lk, [&]() { return S.pending == 0 || abort.load(); });
        }
        if (abort.load()) return;
        {
            std::unique_lock<std::mutex> lk(laneMx);
            laneCv.wait(lk, [&]() { return !freeLanes.empty() || abort.load(); });
            if (freeLanes.empty()) return;                           // (the device could not be made: the batch is given up)
            // the free lane whose staging buffer fits best (the smallest that is large enough, else the largest): a buffer that
            // must grow in the middle of a batch costs a pinned allocation between the slice's front end and its upload
            size_t pick = freeLanes.size() - 1;
            if (sliceEst[si]) {
                const uint64_t need = sliceEst[si] - sliceEst[si] / 16;
                bool fits = lanes[freeLanes[pick]]->capStage >= need;
                for (size_t i = freeLanes.size(); i-- > 0;) {
                    const size_t cap = lanes[freeLanes[i]]->capStage, best = lanes[freeLanes[pick]]->capStage;
                    if (cap >= need ? (!fits || cap < best) : (!fits && cap > best)) { pick = i; fits = cap >= need; }
                }
            }
            S.lane = (int)freeLanes[pick]; freeLanes.erase(freeLanes.begin() + (ptrdiff_t)pick);
            ++slicesWithLane;
        }
        S.tReady = nowMs();
        try {
            const uint32_t first = cut[si], count = cut[si + 1] - cut[si];
            S.plans.resize(count);
            S.items.reserve((size_t)count * S_PE_COUNT);
            uint64_t inBytes = 0;
            // device-side quality path: per bin the place of its packed scores in the input; the gathered streams' places
            std::vector<uint64_t> packedOff(count, 0), nFirst(count, 0); std::vector<uint32_t> gatherItems; uint64_t gatherBytes = 0, nStrings = 0, gatherSymbols = 0, nListBytes = 0;
            uint32_t gatherBits = 0;
            std::vector<uint32_t> idItems;
            // --lossy libraries: one read-only model blob per library in front of the streams
            std::vector<uint64_t> qvzOff(archives.size(), ~0ull);
            // device-side read-id tokeniser: one field table per library in the input, per bin its packed headers, its job
            std::vector<uint64_t> idTabOff(archives.size(), ~0ull); std::vector<std::vector<uint8_t>> idTab(archives.size());
            std::vector<uint32_t> idTok(archives.size(), 0), idVal(archives.size(), 0);
            std::vector<uint64_t> headOff(count, 0); std::vector<fsdev::IdJob> idJobs; std::vector<uint32_t> idJobBin; uint64_t idBytes = 0, nIdStrings = 0;
            // device-side emission (fsdev::EmitOp): per bin a job -- its ops, LZ ids, bases and contig bytes in the input, the streams it writes
            std::vector<fsdev::EmitJob> emitJobs; std::vector<uint32_t> emitJobBin, emitItems; uint64_t emitBytes = 0, nEmitOps = 0, nEmitIds = 0;
            auto emitChannelOf = [](uint32_t s) -> int {
                switch (s) {
                case S_HardReads: return fsdev::ECH_HARD; case S_LettersX: return fsdev::ECH_LETTERS; case S_Match: return fsdev::ECH_MATCH_BITS; case S_MatchBinary: return fsdev::ECH_MATCH_BIN;
                case S_CMatch: return fsdev::ECH_CMATCH_BITS; case S_CLetters: return fsdev::ECH_CLETTERS; case S_LzId: return fsdev::ECH_COUNT;
                case S_HardPE: return fsdev::ECH_HARD_PE; case S_LettersXPE: return fsdev::ECH_LETTERS_PE; case S_MatchRlePE: return fsdev::ECH_MATCH_BITS_PE; case S_MatchBinaryPE: return fsdev::ECH_MATCH_BIN_PE;
                default: return -1;
                }
            };
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t b = byWork[first + k], a = binArch[b];
                if (st[b].idRefs.empty() || idTabOff[a] != ~0ull) continue;
                idFieldBlob(archives[a].head, idTab[a], idTok[a], idVal[a]);
                idTabOff[a] = inBytes; inBytes += (idTab[a].size() + 15) & ~15ull;
            }
            // ... and, when the device symbolises their scores (fs_gather_quality_qvz), the quantizer tables and as many outputs of the
            // archive's generator as the slice's longest stream of that library draws from (four 7-bit draws per output)
            std::vector<uint64_t> qvzSymOff(archives.size(), ~0ull); std::vector<uint32_t> qvzWell(archives.size(), 0);
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t b = byWork[first + k], a = binArch[b];
                if (archives[a].cfg.quaParams.method != MET_QVZ) continue;
                if (!st[b].quaRefs.empty()) qvzWell[a] = std::max<uint32_t>(qvzWell[a], (uint32_t)(st[b].quaSymbols / 4u + 2u));
                if (qvzOff[a] != ~0ull) continue;
                if (!archives[a].qvz.present) throw std::runtime_error("QVZ archive without its codebook (fsgpu_set_quality_codebook)");
                qvzOff[a] = inBytes; inBytes += (archives[a].qvz.blob.size() + 15) & ~15ull;
            }
            for (size_t a = 0; a < archives.size(); ++a) if (qvzWell[a]) { qvzSymOff[a] = inBytes; inBytes += (archives[a].qvz.symBlobBytes(qvzWell[a]) + 15u) & ~15ull; }
            bool gatherQvz = false;
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t b = byWork[first + k];
                const BinIn& bin = info[b]; BinStreams& bs = st[b];
                const BinModuleConfigRaw& binCfg = archives[binArch[b]].cfg;
                const uint32_t qm = binCfg.quaParams.method;
                BlockPlan& pl = S.plans[k]; memset(&pl, 0, sizeof pl);
                pl.signature = bin.signature; pl.records = bin.recCount; pl.raw_dna_size = bin.rawDnaSize; pl.raw_id_size = bs.rawIdSize;
                pl.min_len = (uint8_t)bin.minLen; pl.max_len = (uint8_t)bin.maxLen; pl.has_headers = binCfg.archiveType.readsHaveHeaders != 0;
                pl.n_streams = bs.nStreams; pl.first_item = (uint32_t)S.items.size();
                uint32_t c = 0;
                for (uint32_t s = 0; s < bs.nStreams; ++s) if (streamIsRangeCoded(s, qm)) pl.copy_order[c++] = s;
                for (uint32_t s = 0; s < bs.nStreams; ++s) if (!streamIsRangeCoded(s, qm)) pl.copy_order[c++] = s;
                for (uint32_t s = 0; s < bs.nStreams; ++s) {
                    const bool rc = streamIsRangeCoded(s, qm);
                    if (s == S_Quality && !bs.quaRefs.empty()) {       // device-side quality path: the stream is gathered on the device
                        if (!bs.quaPacked) throw std::runtime_error("quality references without packed scores");
                        const bool qvzQ = qm == MET_QVZ;
                        const uint32_t bits = (qm == MET_NONE || qvzQ) ? 6u : (qm == MET_8BIN ? 3u : 1u);
                        if (gatherBits && (gatherBits != bits || gatherQvz != qvzQ)) throw std::runtime_error("libraries of different quality modes in one device batch");
                        gatherBits = bits; gatherQvz = qvzQ;
                        if (bs.quaSymbols > (qvzQ ? 0x38000000ull : 0x70000000ull)) throw std::runtime_error("stream larger than 4 GiB");
                        StreamItem it; memset(&it, 0, sizeof it);
                        it.bin = k;
                        const uint64_t outBytes = qvzQ ? 4ull * bs.quaSymbols : (rc ? 2ull * bs.quaSymbols : bs.quaSymbols);      // (context, state) words for the QVZ coder, (symbol, context) pairs for the range coder, bytes for PPMd
                        if (qvzQ) { it.kind = KIND_QVZ; it.in_len = (uint32_t)bs.quaSymbols; it.out_cap = 3 * it.in_len + 64; it.aux_off = qvzOff[binArch[b]]; pl.work_size[s] = ~0ull; }
                        else if (rc) { it.kind = KIND_RC_BASE + streamModel(s, qm); it.in_len = (uint32_t)bs.quaSymbols; it.out_cap = 2 * it.in_len + 32; pl.work_size[s] = ~0ull; }
                        else { it.kind = KIND_PPMD; it.in_len = (uint32_t)bs.quaSymbols; it.out_cap = (uint32_t)(bs.quaSymbols + bs.quaSymbols / 8 + 64); pl.work_size[s] = bs.quaSymbols; }
                        it.in_off = gatherBytes;                         // relative to the gather region for now (its base is known after the layout)
                        gatherBytes += (outBytes + 15u + 16u) & ~15ull;
                        gatherItems.push_back((uint32_t)S.items.size());
                        S.items.push_back(it);
                        packedOff[k] = inBytes; inBytes += (bs.quaPackedBytes + 8u + 15u) & ~15ull;
                        nFirst[k] = nListBytes; nListBytes += bs.quaN.size();
                        nStrings += bs.quaRefs.size(); gatherSymbols += bs.quaSymbols;
                        continue;
                    }
                    const uint64_t bytes = bs.s[s].size();
                    if (bytes > 0xF0000000ull) throw std::runtime_error("stream larger than 4 GiB");
                    if ((s == S_IdToken || s == S_IdValue) && !bs.idRefs.empty()) {      // tokenised on the device: in_len is set there (here: its bound)
                        const uint32_t a = binArch[b];
                        uint32_t model = streamModel(s, qm); if (model == 5 && archives[a].head.fields.size() <= 16) model = 6;
                        StreamItem it; memset(&it, 0, sizeof it);
                        const uint64_t pairs = (uint64_t)bs.idRefs.size() * (s == S_IdToken ? idTok[a] : idVal[a]);
                        if (pairs > 0x70000000ull) throw std::runtime_error("stream larger than 4 GiB");
                        it.bin = k; it.kind = KIND_RC_BASE + model; it.in_len = (uint32_t)pairs; it.out_cap = 2 * it.in_len + 32; pl.work_size[s] = ~0ull;
                        it.in_off = idBytes;                             // relative to the read-id part of the device-only region for now
                        if (s == S_IdToken) {
                            fsdev::IdJob j; memset(&j, 0, sizeof j);
                            j.first = (uint32_t)nIdStrings; j.count = (uint32_t)bs.idRefs.size(); j.tok_item = (uint32_t)S.items.size(); j.tok_out = idBytes; j.table_off = idTabOff[a];
                            idJobs.push_back(j); idJobBin.push_back(k);
                            nIdStrings += bs.idRefs.size();
                            headOff[k] = inBytes; inBytes += (bs.headPackedBytes + 8u + 15u) & ~15ull;
                        } else { idJobs.back().val_item = (uint32_t)S.items.size(); idJobs.back().val_out = idBytes; }
                        idBytes += (2 * pairs + 15u + 16u) & ~15ull;
                        idItems.push_back((uint32_t)S.items.size());
                        S.items.push_back(it);
                        continue;
                    }
                    if (bs.deviceEmit && emitChannelOf(s) >= 0) {      // written on the device from the walk's ops: in_len is set there (here: its bound)
                        const uint32_t c = (uint32_t)emitChannelOf(s);
                        if (emitJobBin.empty() || emitJobBin.back() != k) {      // (the bin's first device-written stream: its job)
                            fsdev::EmitJob j; memset(&j, 0, sizeof j);
                            for (uint32_t q = 0; q <= fsdev::ECH_COUNT; ++q) j.item[q] = 0xFFFFFFFFu;
                            j.first_op = (uint32_t)nEmitOps; j.n_ops = (uint32_t)bs.emitOps.size(); j.first_id = (uint32_t)nEmitIds; j.n_ids = (uint32_t)bs.lzIds.size();
                            nEmitOps += bs.emitOps.size(); nEmitIds += bs.lzIds.size();
                            emitJobs.push_back(j); emitJobBin.push_back(k);
                        }
                        fsdev::EmitJob& j = emitJobs.back();
                        const uint64_t bound = bs.emitBound[c] + 2u;
                        if (bound > 0x38000000ull) throw std::runtime_error("stream larger than 4 GiB");
                        StreamItem it; memset(&it, 0, sizeof it);
                        it.bin = k; it.in_len = (uint32_t)bound;
                        const bool bits = c < fsdev::ECH_COUNT && (c == fsdev::ECH_MATCH_BITS || c == fsdev::ECH_CMATCH_BITS || c == fsdev::ECH_MATCH_BITS_PE);
                        const uint64_t unit = (c == fsdev::ECH_COUNT || bits || c == fsdev::ECH_HARD || c == fsdev::ECH_HARD_PE) ? 1u : 2u;
                        if (rc) { uint32_t model = streamModel(s, qm); it.kind = KIND_RC_BASE + model; it.out_cap = 2 * it.in_len + 32; pl.work_size[s] = ~0ull; }
                        else { it.kind = KIND_PPMD; it.out_cap = (uint32_t)(bound + bound / 8 + 64); pl.work_size[s] = ~1ull; }      // (~1: the length the kernels leave in the item)
                        j.item[c] = (uint32_t)S.items.size(); j.cap[c] = (uint32_t)bound;
                        j.out_off[c] = emitBytes; it.in_off = emitBytes;            // relative to the emission's part of the device-only region for now
                        emitBytes += (unit * bound + 2u + 15u) & ~15ull;
                        if (bits) { j.raw_off[c] = emitBytes; emitBytes += (bound + 15u) & ~15ull; }
                        emitItems.push_back((uint32_t)S.items.size());
                        S.items.push_back(it);
                        continue;
                    }
                    StreamItem it; memset(&it, 0, sizeof it);
                    it.bin = k; it.in_off = inBytes;
                    // header-less archives never create the read-id coders: their streams stay empty (FastqCompressor.cpp:923-930)
                    const bool absent = !pl.has_headers && (s == S_IdToken || s == S_IdValue);
                    if (absent) { it.kind = KIND_PPMD; it.in_len = 0; it.out_cap = 16; pl.work_size[s] = 0; }
                    else if (rc && s == S_Quality && qm == MET_QVZ) {
                        it.kind = KIND_QVZ; it.in_len = (uint32_t)(bytes / 4); it.out_cap = 3 * it.in_len + 64; it.aux_off = qvzOff[binArch[b]]; pl.work_size[s] = ~0ull; }
                    else if (rc) { uint32_t model = streamModel(s, qm);
                                   if (model == 5 && archives[binArch[b]].head.fields.size() <= 16) model = 6;     // read-id ctx0 = fieldId*4+k < 64: dense 8 MiB table
                                   it.kind = KIND_RC_BASE + model; it.in_len = (uint32_t)(bytes / 2); it.out_cap = 2 * it.in_len + 32; pl.work_size[s] = ~0ull; }
                    else { it.kind = KIND_PPMD; it.in_len = (uint32_t)bytes; it.out_cap = (uint32_t)(bytes + bytes / 8 + 64); pl.work_size[s] = bytes; }
                    S.items.push_back(it);
                    inBytes += (bytes + 15) & ~15ull;
                }
            }
            fsdev::GatherPlan gp; uint64_t gatherBaseQ = 0;
            uint64_t devBase = 0;          // where the device-only region stands as the items placed in it so far see it (the input grows with every plan behind them)
            if (nStrings) {
                if (nStrings > 0xFFFFFFF0ull || gatherBytes > 0xF0000000ull) throw std::runtime_error("quality gather larger than 4 GiB");
                gp.desc_off = inBytes; gp.n_strings = (uint32_t)nStrings; gp.out_bytes = gatherBytes; gp.symbols = gatherSymbols; gp.bits = gatherBits; gp.qvz = gatherQvz ? 1u : 0u;
                inBytes += (nStrings * (gatherQvz ? sizeof(fsdev::QuaQvzString) : (gatherBits == 6u ? sizeof(fsdev::QuaString) : sizeof(fsdev::QuaPairString))) + 15u) & ~15ull;
                if (gatherBits != 6u) {
                    if (nListBytes > 0xFFFFFFF0ull) throw std::runtime_error("quality gather larger than 4 GiB");
                    gp.n_list_off = inBytes; gp.n_list_bytes = nListBytes; inBytes += (nListBytes + 15u) & ~15ull;
                    // a stored 0 / 1 stands for the scores 6 / 40 (FastqPacker); the coded symbol is the score against the archive's threshold
                    const uint32_t thr = archives[binArch[byWork[first]]].cfg.quaParams.binaryThreshold;
                    gp.sym_of_bit[0] = 6u >= thr ? 1u : 0u; gp.sym_of_bit[1] = 40u >= thr ? 1u : 0u;
                }
                gatherBaseQ = (inBytes + 15u) & ~15ull;
                for (uint32_t gi : gatherItems) S.items[gi].in_off += gatherBaseQ;
                devBase = gatherBaseQ;
            }
            fsdev::IdPlan ip;
            if (!idJobs.empty()) {
                if (nIdStrings > 0xFFFFFFF0ull || gatherBytes + idBytes > 0xF0000000ull) throw std::runtime_error("read-id streams larger than 4 GiB");
                ip.n_jobs = (uint32_t)idJobs.size(); ip.n_strings = (uint32_t)nIdStrings; ip.out_bytes = idBytes;
                ip.jobs_off = inBytes; inBytes += (idJobs.size() * sizeof(fsdev::IdJob) + 15u) & ~15ull;
                ip.strings_off = inBytes; inBytes += (nIdStrings * sizeof(fsdev::IdString) + 15u) & ~15ull;
                const uint64_t gatherBase = (inBytes + 15u) & ~15ull;
                // (the quality part's items were based on an input that has grown since: base them again)
                if (nStrings) for (uint32_t gi : gatherItems) S.items[gi].in_off += gatherBase - gatherBaseQ;
                for (uint32_t gi : idItems) S.items[gi].in_off += gatherBase + gatherBytes;
                for (fsdev::IdJob& j : idJobs) { j.tok_out += gatherBytes; j.val_out += gatherBytes; }
                devBase = gatherBase;
            }
            fsdev::EmitPlan ep;
            std::vector<uint64_t> emitSeqOff(emitJobs.size(), 0), emitContigOff(emitJobs.size(), 0);
            if (!emitJobs.empty()) {
                if (nEmitOps > 0xFFFFFFF0ull || emitBytes > 0xF0000000ull) throw std::runtime_error("emitted streams larger than 4 GiB");
                for (size_t j = 0; j < emitJobs.size(); ++j) {
                    const BinStreams& bs = st[byWork[first + emitJobBin[j]]];
                    emitSeqOff[j] = inBytes; inBytes += (bs.emitSeqHi - bs.emitSeqLo + 15u) & ~15ull;
                    emitContigOff[j] = inBytes; inBytes += (bs.contigBytes.size() + 15u) & ~15ull;
                }
                ep.n_jobs = (uint32_t)emitJobs.size(); ep.n_ops = (uint32_t)nEmitOps; ep.n_ids = (uint32_t)nEmitIds; ep.out_bytes = emitBytes;
                ep.jobs_off = inBytes; inBytes += (emitJobs.size() * sizeof(fsdev::EmitJob) + 15u) & ~15ull;
                ep.ops_off = inBytes; inBytes += (nEmitOps * sizeof(fsdev::EmitOp) + 15u) & ~15ull;
                ep.ids_off = inBytes; inBytes += (4ull * nEmitIds + 15u) & ~15ull;
                // (the device-only region begins behind an input that has grown: the earlier parts' items are based again, this part's for the first time)
                const uint64_t newBase = (inBytes + 15u) & ~15ull;
                if (nStrings) for (uint32_t gi : gatherItems) S.items[gi].in_off += newBase - devBase;
                for (uint32_t gi : idItems) S.items[gi].in_off += newBase - devBase;
                for (uint32_t gi : emitItems) S.items[gi].in_off += newBase + gatherBytes + idBytes;
                devBase = newBase;
                for (size_t j = 0; j < emitJobs.size(); ++j) {
                    const uint32_t b = byWork[first + emitJobBin[j]]; const BinStreams& bs = st[b];
                    const BinModuleConfigRaw& binCfg = archives[binArch[b]].cfg;
                    fsdev::EmitJob& jb = emitJobs[j];
                    jb.seq_off = emitSeqOff[j]; jb.seq_bytes = (uint32_t)(bs.emitSeqHi - bs.emitSeqLo); jb.contig_off = emitContigOff[j]; jb.contig_bytes = (uint32_t)bs.contigBytes.size();
                    jb.sig_len = binCfg.minimizer.signatureLen; jb.begin_cut = par.beginCut; jb.end_cut = par.endCut;
                    memset(jb.dna_to_idx, 0xFF, sizeof jb.dna_to_idx);
                    for (int q = 0; q < 5; ++q) jb.dna_to_idx[(uint8_t)binCfg.minimizer.dnaSymbolOrder[q] & 127u] = (uint8_t)q;
                }
            }
            fsengine::Device* L;
            { std::lock_guard<std::mutex> lk(laneMx); L = lanes[(uint32_t)S.lane]; }
            const double tb = nowMs();
            uint8_t* input = fsengine::staging_buffer(L, inBytes + 16);        // pinned host memory owned by the lane
            S.bufMs = nowMs() - tb;
            if (!input) throw std::runtime_error(std::string("device: ") + L->err);
            for (size_t a = 0; a < archives.size(); ++a) if (qvzOff[a] != ~0ull) memcpy(input + qvzOff[a], archives[a].qvz.blob.data(), archives[a].qvz.blob.size());
            for (size_t a = 0; a < archives.size(); ++a) if (qvzSymOff[a] != ~0ull) archives[a].qvz.writeSymBlob(input + qvzSymOff[a], qvzWell[a]);
            for (size_t a = 0; a < archives.size(); ++a) if (idTabOff[a] != ~0ull) memcpy(input + idTabOff[a], idTab[a].data(), idTab[a].size());
            if (!idJobs.empty()) {
                memcpy(input + ip.jobs_off, idJobs.data(), idJobs.size() * sizeof(fsdev::IdJob));
                parallelFor((uint32_t)idJobs.size(), 4, [&](uint32_t j, uint32_t) {
                    const uint32_t k = idJobBin[j], b = byWork[first + k];
                    memcpy(input + headOff[k], st[b].headPacked, st[b].headPackedBytes);
                    memset(input + headOff[k] + st[b].headPackedBytes, 0, 8);
                    fsdev::IdString* ss = (fsdev::IdString*)(input + ip.strings_off) + idJobs[j].first;
                    const uint64_t maxBit = 8ull * st[b].headPackedBytes;
                    for (const IdRef& r : st[b].idRefs) {
                        if (r.len > 1u && (uint64_t)r.bit + 7ull * (r.len - 1u) > maxBit) throw std::runtime_error("Corrupted bin: read id outside the bin's header bytes");
                        ss->src_bit = 8ull * headOff[k] + r.bit; ss->len = r.len; ss->pad = 0; ++ss;
                    }
                });
            }
            if (!emitJobs.empty()) {
                memcpy(input + ep.jobs_off, emitJobs.data(), emitJobs.size() * sizeof(fsdev::EmitJob));
                parallelFor((uint32_t)emitJobs.size(), 4, [&](uint32_t j, uint32_t) {
                    const BinStreams& bs = st[byWork[first + emitJobBin[j]]];
                    memcpy(input + emitSeqOff[j], bs.emitSeq, bs.emitSeqHi - bs.emitSeqLo);
                    if (!bs.contigBytes.empty()) memcpy(input + emitContigOff[j], bs.contigBytes.data(), bs.contigBytes.size());
                    fsdev::EmitOp* ops = (fsdev::EmitOp*)(input + ep.ops_off) + emitJobs[j].first_op;
                    if (!bs.emitOps.empty()) memcpy(ops, bs.emitOps.data(), bs.emitOps.size() * sizeof(fsdev::EmitOp));
                    for (size_t q = 0; q < bs.emitOps.size(); ++q) ops[q].pad2[0] = j;
                    if (!bs.lzIds.empty()) memcpy(input + ep.ids_off + 4ull * emitJobs[j].first_id, bs.lzIds.data(), 4ull * bs.lzIds.size());
                });
            }
            // staging copy on a few helper threads of its own (the host threads are busy with the next slices' front end)
            std::vector<uint64_t> stringBase(count + 1, 0);          // first descriptor of every bin
            if (nStrings) for (uint32_t k = 0; k < count; ++k) stringBase[k + 1] = stringBase[k] + st[byWork[first + k]].quaRefs.size();
            parallelFor(count, 4, [&](uint32_t k, uint32_t) {
                const BlockPlan& pl = S.plans[k]; const uint32_t b = byWork[first + k];
                const bool gathered = !st[b].quaRefs.empty();
                for (uint32_t s = 0; s < pl.n_streams; ++s) {
                    if (gathered && s == S_Quality) continue;
                    if (st[b].deviceEmit && emitChannelOf(s) >= 0) continue;
                    const auto& v = st[b].s[s];
                    if (!v.empty()) memcpy(input + S.items[pl.first_item + s].in_off, v.data(), v.size());
                }
                if (gathered) {
                    memcpy(input + packedOff[k], st[b].quaPacked, st[b].quaPackedBytes);
                    memset(input + packedOff[k] + st[b].quaPackedBytes, 0, 8);
                    // the stream's place in the gather region, relative to its base (in_off is absolute: the base is the
                    // input size rounded up to 16, see encode_batch)
                    uint64_t dst = S.items[pl.first_item + S_Quality].in_off - ((inBytes + 15u) & ~15ull);
                    const uint64_t srcBase = 8ull * packedOff[k], maxBit = 8ull * st[b].quaPackedBytes;
                    if (gp.qvz) {
                        fsdev::QuaQvzString* qs = (fsdev::QuaQvzString*)(input + gp.desc_off) + stringBase[k];
                        const uint32_t model16 = (uint32_t)(qvzSymOff[binArch[b]] >> 4);
                        uint32_t draw = 0;
                        for (const QuaRef& r : st[b].quaRefs) {
                            if ((uint64_t)r.bit + 6ull * r.len > maxBit) throw std::runtime_error("Corrupted bin: quality string outside the bin's quality bytes");
                            qs->src_bit = srcBase + r.bit; qs->dst_off = (uint32_t)dst; qs->draw0 = draw; qs->model16 = model16; qs->len = r.len; qs->reverse = r.reverse; ++qs;
                            dst += 4ull * r.len; draw += r.len;
                        }
                    } else if (gp.bits == 6u) {
                        fsdev::QuaString* qs = (fsdev::QuaString*)(input + gp.desc_off) + stringBase[k];
                        for (const QuaRef& r : st[b].quaRefs) {
                            if ((uint64_t)r.bit + 6ull * r.len > maxBit) throw std::runtime_error("Corrupted bin: quality string outside the bin's quality bytes");
                            qs->src_bit = srcBase + r.bit; qs->dst_off = (uint32_t)dst; qs->len = r.len; qs->reverse = r.reverse; ++qs;
                            dst += r.len;
                        }
                    } else {
                        fsdev::QuaPairString* qs = (fsdev::QuaPairString*)(input + gp.desc_off) + stringBase[k];
                        if (!st[b].quaN.empty()) memcpy(input + gp.n_list_off + nFirst[k], st[b].quaN.data(), st[b].quaN.size());
                        dst /= 2;                                       // pairs
                        for (const QuaRef& r : st[b].quaRefs) {
                            if ((uint64_t)r.bit + (uint64_t)gp.bits * r.len > maxBit) throw std::runtime_error("Corrupted bin: quality string outside the bin's quality bytes");
                            qs->src_bit = srcBase + r.bit; qs->dst_off = (uint32_t)dst; qs->n_off = (uint32_t)(nFirst[k] + r.nFirst); qs->len = r.len; qs->reverse = r.reverse; qs->n_count = r.nCount; ++qs;
                            dst += r.len - r.nCount;
                        }
                    }
                }
            });
            S.tSubmit = nowMs(); S.inBytes = inBytes;
            // (everything the slice needs of the batch's arrays is in its staging buffer now)
            if (++slicesStaged == nSlices && releaseEarly && onAllStaged) { std::lock_guard<std::mutex> g(releaserMx); stagedReleaser = std::thread(onAllStaged); }
            if (fsengine::encode_batch(L, input, inBytes, S.items, S.plans, sliceBlocks[si], S.sizes, &S.timing, nStrings ? &gp : nullptr, idJobs.empty() ? nullptr : &ip, emitJobs.empty() ? nullptr : &ep) != 0) S.err = std::string("device: ") + L->err;
        } catch (const std::exception& e) { S.err = e.what(); }
        S.tDone = nowMs();
        // (a context that packs once: no further slice will ask for a lane -- this one's staging buffer goes now, beside the device's tail)
        if (releaseEarly && slicesWithLane.load() == nSlices && S.lane >= 0) fsengine::staging_release(lanes[(uint32_t)S.lane]);
        { std::lock_guard<std::mutex> lk(laneMx); freeLanes.push_back((uint32_t)S.lane); }
        laneCv.notify_one();
        { std::lock_guard<std::mutex> lk(S.mx); S.done = true; }
        S.cv.notify_all();
    };
    for (uint32_t si = 0; si < nSlices; ++si) slices[si].th = std::thread(runSlice, si);
    // FS_WATCHDOG=<seconds> (diagnostic): a batch that has not finished by then reports where it stands and ends the process
    std::mutex wdMx; std::condition_variable wdCv; bool wdStop = false; std::thread watchdog;
    if (const char* wd = getenv("FS_WATCHDOG")) {
        const int secs = atoi(wd);
        if (secs > 0) watchdog = std::thread([&, secs]() {
            std::unique_lock<std::mutex> lk(wdMx);
            if (wdCv.wait_for(lk, std::chrono::seconds(secs), [&]() { return wdStop; })) return;
            fprintf(stderr, "[watchdog] batch of %u bins in %u slices on %u lanes not finished after %d s\n", nBins, nSlices, nLanes, secs);
            for (uint32_t si = 0; si < nSlices; ++si) {
                Slice& S = slices[si];
                uint32_t pend; bool dn; { std::lock_guard<std::mutex> g(S.mx); pend = S.pending; dn = S.done; }
                fprintf(stderr, "[watchdog] slice %u: bins pending %u, ready %.0f ms, submitted %.0f ms, done %d, streams %zu\n", si, pend, S.tReady, S.tSubmit, (int)dn, S.items.size());
            }
            fflush(stderr);
            for (uint32_t si = 0; si < nSlices; ++si) {
                Slice& S = slices[si];
                bool dn; { std::lock_guard<std::mutex> g(S.mx); dn = S.done; }
                if (dn || S.tSubmit <= 0 || S.lane < 0) continue;
                fprintf(stderr, "[watchdog] slice %u device state: ", si); fflush(stderr);
                char buf[1536] = {0};
                fsengine::lane_debug(lanes[(uint32_t)S.lane], buf, sizeof buf);
                fprintf(stderr, "%s\n", buf); fflush(stderr);
            }
            _exit(86);
        });
    }
    auto joinAll = [&]() {
        if (laneMaker.joinable()) laneMaker.join();
        for (Slice& s : slices) if (s.th.joinable()) s.th.join();
        { std::lock_guard<std::mutex> g(releaserMx); if (stagedReleaser.joinable()) stagedReleaser.join(); }
        nLanes = madeLanes.load();
        if (watchdog.joinable()) { { std::lock_guard<std::mutex> g(wdMx); wdStop = true; } wdCv.notify_all(); watchdog.join(); }
    };

    std::vector<double> busyMs(hostThreads, 0.0);
    std::vector<std::unique_ptr<BinEncoder>>& encs = encoders;        // kept across calls: their work buffers stay mapped
    if (encs.size() < hostThreads) encs.resize(hostThreads);
    if (matchLanes.size() < hostThreads) matchLanes.resize(hostThreads, nullptr);
    uint32_t matcherBins = deviceMatcher ? 6u * hostThreads : 0u;      // measured on the BASELINE library: 72 bins 1841, 144 bins 1923, 250+ bins below 1750 MB/s (the searches start to wait for registers)
    if (const char* mb = getenv("FS_MATCHER_BINS")) matcherBins = (uint32_t)std::max(0, atoi(mb));
    // (the lighter bins' searches: how many threads may wait for the device at once -- the threads the process has beyond its cores, in
    // a paired-end batch, whose step is the host's; a single-end step is its longest stream on the device, and searches that run
    // beside it cost it 4 %: profiles/r05_search_surplus.txt)
    bool pairedBatch = false;
    for (uint32_t b = 0; b < nBins && !pairedBatch; ++b) pairedBatch = archives[binArch[b]].cfg.archiveType.readType == READ_PE;
    uint32_t searchSurplus = pairedBatch && hostCores && hostThreads > hostCores ? hostThreads - hostCores : 0u;
    if (const char* ss = getenv("FS_SEARCH_SURPLUS")) searchSurplus = (uint32_t)std::max(0, atoi(ss));
    if (deviceMatcher && matchReserve.reads && matcherBins) {
        // the matcher lanes of an earlier batch get room for this batch's largest bin now, while no coder kernel is in
        // flight (growing frees, and a free waits for every running kernel); lanes that do not exist yet are made by
        // their threads (matcherFor)
        for (uint32_t t = 0; t < hostThreads; ++t)
            if (matchLanes[t] && fsengine::match_lane_reserve(device(), matchLanes[t], matchReserve.reads, matchReserve.seqBytes, matchReserve.calls, matchReserve.warm) != 0) throw std::runtime_error(std::string("device: ") + device()->err);
    }
    const double tf = nowMs();
    if (trace) fprintf(stderr, "[trace] batch set-up (slices, %u lanes, matcher lanes) %.1f ms\n", nLanes, tf - t0);
    try {
        std::mutex roundMx; std::condition_variable roundCv; uint32_t binsDone = 0;
        // The mate searches of paired-end bins on the device (fs_match_mates), batched and not waited for: FS_DEVICE_MATES=2.  NOT the default: a bin's
        // pairs are a chain -- 11 us a pair in a 1024-thread workgroup that wants a compute unit to itself, against ~1.6 us a pair and thread on the
        // host -- and the coder kernels hold the compute units for most of a step: round 5, 6 M pairs: 2.19 s a step with the host's search, 2.92 s
        // with the device's for every bin, 3.4 s with the device's for all but the heaviest bins (profiles/r05_mate_search_modes.txt).  The kernel
        // stays under test row for row and archive for archive.
        const int matesMode = getenv("FS_DEVICE_MATES") ? atoi(getenv("FS_DEVICE_MATES")) : 0;       // (read per batch: the tests switch it inside one process)
        const bool deviceEmit = !(getenv("FS_DEVICE_EMIT") && atoi(getenv("FS_DEVICE_EMIT")) == 0);
        std::mutex asyncErrMx; std::string asyncErr;
        // (a bin whose pairs are with the device is complete when BOTH its host task has returned -- it still fills in the bin's packed
        // scores and read ids behind the walk -- and its rows have come back: whichever is second counts it)
        // A slice lives in ONE 32-bit address space on the device: what goes up (packed scores and read ids, descriptors, the small streams) and
        // what the device writes behind it (the gathered quality stream, a byte a score; the read-id pairs) -- ~3.5 bytes a base --, and for the
        // bins whose base-holding streams the device writes itself their bases, ops and the ROOM of everything those ops may write: ~10 bytes
        // a base in all.  Slices are cut by weight, so with few slices (pipeline_slices <= 4, or fewer than 64 bins: one slice), or with the
        // bins of a 25 M-pair library (the twelve heaviest hold 0.6 GB of bases: found by the round's first 25 M-pair run, "quality gather plan
        // outside the batch input"), a slice would pass 4 GiB.  Every bin claims its footprint from its slice's budget before its walk starts;
        // a bin that finds the budget used up keeps the walk's own streams (same bytes in the archive).
        std::unique_ptr<std::atomic<uint64_t>[]> sliceDevBytes(new std::atomic<uint64_t>[nSlices]);
        for (uint32_t si = 0; si < nSlices; ++si) sliceDevBytes[si].store(0);
        const std::vector<uint64_t> basesOf = binBases.size() == nBins ? binBases : std::vector<uint64_t>();
        binBases.clear();
        static const uint64_t sliceDevCap = getenv("FS_SLICE_DEV_CAP") ? (uint64_t)atoll(getenv("FS_SLICE_DEV_CAP")) : (3700ull << 20);      // (the variable: tests of the fall-back)
        std::unique_ptr<std::atomic<uint8_t>[]> parts(new std::atomic<uint8_t>[nBins]);
        for (uint32_t i = 0; i < nBins; ++i) parts[i].store(2);
        // a bin is through the front end: its slice counts it (here, or -- its pairs still with the device -- when they come back)
        auto binComplete = [&](uint32_t k) {
            Slice& S = slices[sliceOf[k]];
            bool last;
            { std::lock_guard<std::mutex> lk(S.mx); last = --S.pending == 0; }
            if (last) S.cv.notify_all();
        };
        // (declared behind everything its workers' callbacks touch -- parts, the error string and its mutex, binComplete --: should a host
        // task throw, the unwinding joins the workers FIRST, while all of that is still there)
        std::unique_ptr<MateDispatcher> mateDispatcher;
        if (matesMode == 2 && deviceEmit && deviceMatcher) mateDispatcher.reset(new MateDispatcher(*this));
        parallelFor(nBins, hostThreads, [&](uint32_t k, uint32_t tid) {
            const uint32_t b = byWork[k];
            if (abort.load()) return;                                // (the device could not be made: nothing left to do for the bins)
            if (k >= firstRound && k < hostThreads) {            // a thread beyond the cores: its first bin starts when a core is free
                std::unique_lock<std::mutex> lk(roundMx);
                roundCv.wait(lk, [&]() { return binsDone > k - firstRound || abort.load(); });
            }
            // (a split pack: the pipelines share the worker slots, the heaviest class first)
            struct Slot { HostGate* g; uint32_t c; Slot(HostGate* x, uint32_t cls) : g(x), c(cls) { if (g) g->acquire(c); } ~Slot() { if (g) g->release(c); } } slot(hostGate, gateClass);
            struct Done { std::mutex& m; std::condition_variable& cv; uint32_t& n; ~Done() { { std::lock_guard<std::mutex> g(m); ++n; } cv.notify_all(); } } done{roundMx, roundCv, binsDone};     // (also when the bin throws)
            static const bool binTrace = getenv("FS_BIN_TRACE") != nullptr;
            if (binTrace && k < 32) fprintf(stderr, "[bin] rank %u (bin %u, thread %u) claimed at %.1f ms of the batch\n", k, b, tid, nowMs() - t0);
            if (!encs[tid]) encs[tid].reset(new BinEncoder(par));
            if (binTrace && k < 32) fprintf(stderr, "[bin] rank %u: encoder ready at %.1f ms of the batch\n", k, nowMs() - t0);
            // The heaviest bins -- the first few rounds of the host threads -- have their window searches done by the device:
            // they sit on the critical path (their quality streams are the longest) and the device is still nearly empty.
            // Later the coder kernels hold the chip's registers (two waves of 180 VGPRs a SIMD leave 144: a 1024-thread search wants four
            // waves of ~70 there) and a search WAITS for coder workgroups to leave.  A waiting thread costs nothing while more threads
            // are runnable than the process has cores, so the lighter bins' searches go to the device as long as fewer threads are
            // inside a search than that surplus, and are the host's scan otherwise -- in paired-end batches (round 5, profiles/
            // r05_matcher_cap_*.txt, r05_search_surplus.txt: every bin on the device whatever the queue: +6 % on the single-end step, nothing
            // on a paired-end one; with the surplus rule +4 % and -4 %).
            encs[tid]->setMatcher(deviceMatcher ? matcherFor(tid) : MatchFn());
            if (k < matcherBins) encs[tid]->setMatcherGate(nullptr);
            else encs[tid]->setMatcherGate([this, searchSurplus]() { return searchesInFlight.load(std::memory_order_relaxed) < (int)searchSurplus; });
            encs[tid]->setMateMatcher(MateFn());
            // the streams that hold bases (HardReads, LettersX, Match, ...: fsdev::EmitOp) and the LZ ids' run-length coding: written
            // by the device from the ops the walk leaves (FS_DEVICE_EMIT=0: by the walk itself, A/B runs)
            bool emitHere = deviceEmit;
            {
                const uint64_t bases = basesOf.empty() ? weight[b] * 320u : basesOf[b];
                std::atomic<uint64_t>& used = sliceDevBytes[sliceOf[k]];
                if (emitHere && used.fetch_add(10u * bases) + 10u * bases > sliceDevCap) { used.fetch_sub(10u * bases - 7u * bases / 2u); emitHere = false; }
                else if (!emitHere) used.fetch_add(7u * bases / 2u);
            }
            encs[tid]->setDeviceEmit(emitHere);
            if (mateDispatcher) {
                MateDispatcher* md = mateDispatcher.get();
                encs[tid]->setAsyncMates([&, k, md](std::unique_ptr<PendingPairs> pp) {
                    pp->done = [&, k](const char* error) {
                        if (error) { { std::lock_guard<std::mutex> g(asyncErrMx); if (asyncErr.empty()) asyncErr = error; } abort.store(true); for (Slice& s : slices) { std::lock_guard<std::mutex> lk(s.mx); s.cv.notify_all(); } }
                        if (parts[k].fetch_sub(1) == 1) binComplete(k);
                    };
                    return md->push(std::move(pp));
                });
            } else encs[tid]->setAsyncMates(AsyncMateFn());
            const double ta = trace ? nowMs() : 0.0;
            produce(b, *encs[tid], st[b], info[b], recBytes[b]);
            if (trace) busyMs[tid] += nowMs() - ta;
            if (!st[b].pairsPending || parts[k].fetch_sub(1) == 1) binComplete(k);
        });
        mateDispatcher.reset();                                  // (waits for the searches still on their way and their bins' completion)
        if (!asyncErr.empty()) throw std::runtime_error(asyncErr);
    } catch (...) {
        abort.store(true);
        for (Slice& s : slices) { std::lock_guard<std::mutex> lk(s.mx); s.cv.notify_all(); }
        joinAll();
        throw;
    }
    const double feMs = nowMs() - tf;
    if (onHostTasksDone) onHostTasksDone();
    joinAll();
    if (!makerErr.empty()) throw std::runtime_error(makerErr);
    for (Slice& S : slices) if (!S.err.empty()) throw std::runtime_error(S.err);
    if (nSlices > 1) {
        // Any lane may get the largest slice of the next batch: the lanes get the buffers of the best-equipped one NOW, behind
        // the batch (nothing in flight) -- a lane that had to grow a buffer in the middle of the next batch would hipFree, which
        // waits for every kernel in flight.  A one-shot context (the CLI) only notes it: it would pay for pinned and device
        // memory it never uses; should it pack again, the lanes are made alike in front of that batch.
        uint64_t mx = 0; for (Slice& S : slices) mx = std::max(mx, S.inBytes);
        equalizeStage = std::max(equalizeStage, mx + 16); equalizeLanes = nLanes;
        if (!cfg.one_shot) equalizeNow();
    }
    stats.frontend_ms += feMs;
    for (uint32_t si = 0; si < nSlices; ++si) {
        Slice& S = slices[si];
        uint64_t off = 0;
        for (uint32_t k = 0; k < cut[si + 1] - cut[si]; ++k) {
            const uint32_t b = byWork[cut[si] + k];
            blockSizes[b] = S.sizes[k]; blockSlice[b] = si; blockOff[b] = off; off += S.sizes[k];
        }
        timing.encode_ms += S.timing.encode_ms; timing.assemble_ms += S.timing.assemble_ms; timing.launches += S.timing.launches; timing.items += S.timing.items;
        timing.ppmd_symbols += S.timing.ppmd_symbols; timing.rc_symbols += S.timing.rc_symbols; timing.restarts += S.timing.restarts; timing.max_restarts = std::max(timing.max_restarts, S.timing.max_restarts); for (int w = 0; w < 16; ++w) timing.win[w] += S.timing.win[w];
        timing.h2d_bytes += S.timing.h2d_bytes; timing.d2h_bytes += S.timing.d2h_bytes;
        timing.gather_ms += S.timing.gather_ms; timing.gather_symbols += S.timing.gather_symbols; timing.gather_bytes += S.timing.gather_bytes; timing.id_strings += S.timing.id_strings; timing.tail_launches += S.timing.tail_launches;
        if (trace) fprintf(stderr, "[trace] slice %u/%u: %u bins, front end done at %.1f ms, staged+submitted at %.1f ms (%.0f MB; staging buffer %.1f ms), device done at %.1f ms (kernel %.1f ms)\n",
                           si + 1, nSlices, cut[si + 1] - cut[si], S.tReady - t0, S.tSubmit - t0, S.inBytes / 1e6, S.bufMs, S.tDone - t0, S.timing.encode_ms);
    }
    if (trace) {
        double sum = 0, mx = 0; for (double v : busyMs) { sum += v; mx = std::max(mx, v); }
        fprintf(stderr, "[trace] batch: %u bins in %u slices on %u lanes, host tasks %.1f ms (%u threads: busy sum %.0f ms, busiest %.0f ms), total %.1f ms\n",
                nBins, nSlices, nLanes, feMs, hostThreads, sum, mx, nowMs() - t0);
        fprintf(stderr, "[trace] device matcher so far: %llu reads, %.0f ms in its calls (summed over the host threads), %.0f ms of kernels\n",
                (unsigned long long)matchedReads.load(), matchUs.load() / 1e3, matchKernelUs.load() / 1e3);
    }
    stats.bins += nBins;
    binInfo.swap(info);
    for (uint32_t b = 0; b < nBins; ++b) { stats.records += binInfo[b].recCount; stats.algorithmic_bytes += blockSizes[b] + recBytes[b]; }
}

// ------------------------------------------------------------------------------------------------
// RawCompressorSE/PE::Compress (fastore_pack/FastqCompressor.cpp:3407-3600, 5426-5441) preceded by the
// un-reverse-complement / un-swap pass of CompressorModule.cpp:136-149, 718-733.
void Context::compressRawBlock(Batch& batch, const ArchiveParams& arch, std::vector<uint8_t>& out) const
{
    const BinModuleConfigRaw& binCfg = arch.cfg; const HeaderStats& head = arch.head;
    const BinIn& bin = batch.bins.at(0);
    const bool pe = binCfg.archiveType.readType == READ_PE;
    const bool hasHeaders = binCfg.archiveType.readsHaveHeaders != 0;
    const uint32_t qm = binCfg.quaParams.method;
    auto comp = [](uint8_t c) -> uint8_t { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 'N': return 'N'; } return 0xFF; };
    for (uint32_t i = bin.recBegin; i < bin.recBegin + bin.recCount; ++i) {
        Rec& r = batch.recs[i];
        uint8_t* s = batch.seq.data() + r.seqOff; uint8_t* q = batch.qua.data() + r.seqOff;
        const uint32_t len = (uint32_t)r.seqLen + r.auxLen;
        if (r.flags & FLAG_REVERSE) {
            for (uint32_t a = 0, b = len - 1; a < b; ++a, --b) { const uint8_t t = comp(s[a]); s[a] = comp(s[b]); s[b] = t; std::swap(q[a], q[b]); }
            if (len & 1) s[len / 2] = comp(s[len / 2]);
            r.flags &= ~FLAG_REVERSE;
        }
        if (pe && (r.flags & FLAG_SWAPPED)) {
            for (uint32_t a = 0; a < r.seqLen; ++a) { std::swap(s[a], s[a + r.seqLen]); std::swap(q[a], q[a + r.seqLen]); }
            r.flags &= ~FLAG_SWAPPED;
        }
        r.minimPos = 0;
    }
    std::vector<uint8_t> tok, val, dna, quaStream;
    uint64_t rawId = 0;
    WellRng well;
    if (qm == MET_QVZ) well.reset(arch.qvz.wellSeed);            // RawCompressorSE::Compress resets the generator too (FastqCompressor.cpp:3704-3714)
    for (uint32_t i = bin.recBegin; i < bin.recBegin + bin.recCount; ++i) {
        const Rec& r = batch.recs[i];
        const uint8_t* s = batch.seq.data() + r.seqOff; const uint8_t* q = batch.qua.data() + r.seqOff;
        if (hasHeaders) { compressReadId(head, batch.head.data() + r.headOff, r.headLen, tok, val); rawId += r.headLen; }
        dna.insert(dna.end(), s, s + r.seqLen + r.auxLen);
        compressReadQuality(binCfg, s, q, r.seqLen, false, quaStream, &arch.qvz, &well);
        if (pe) compressReadQuality(binCfg, s + r.seqLen, q + r.seqLen, r.auxLen, false, quaStream, &arch.qvz, &well);
    }
    std::vector<uint8_t> cTok, cVal, cDna, cQua;
    std::thread tq([&]() { if (qm == MET_NONE) fshost::ppmdEncode(quaStream.data(), quaStream.size(), cQua);
                           else if (qm == MET_QVZ) fshost::qvzEncode(arch.qvz.blob.data(), quaStream.data(), quaStream.size() / 4, cQua);
                           else fshost::rcEncode(streamModel(S_Quality, qm), quaStream.data(), quaStream.size() / 2, cQua); });
    std::thread ti([&]() { if (hasHeaders) { fshost::rcEncode(5, tok.data(), tok.size() / 2, cTok); fshost::rcEncode(5, val.data(), val.size() / 2, cVal); } });
    fshost::ppmdEncode(dna.data(), dna.size(), cDna);
    tq.join(); ti.join();
    ByteWriter w;
    w.put4(bin.signature); w.put8(bin.recCount); w.put(bin.minLen); w.put(bin.maxLen);
    w.put8(bin.rawDnaSize);
    const uint64_t footerOff = 74 + cTok.size() + cVal.size() + cDna.size() + cQua.size();
    w.put8(footerOff); w.put4(1);
    if (hasHeaders) w.put8(rawId);
    w.put8(cDna.size()); w.put8(cQua.size());
    if (hasHeaders) { w.put8(cTok.size()); w.put8(cVal.size()); }
    while (w.size() < 74) w.put(0);                         // RawBlockHeader::Size, zero-filled (FastqCompressor.cpp:3496)
    w.putBytes(cTok.data(), cTok.size()); w.putBytes(cVal.data(), cVal.size());
    w.putBytes(cDna.data(), cDna.size()); w.putBytes(cQua.data(), cQua.size());
    w.put(0);                                               // footer
    out = std::move(w.b);
}

// ------------------------------------------------------------------------------------------------
std::vector<uint32_t> splitClasses(const std::vector<uint64_t>& records, const std::vector<uint64_t>& bases, uint32_t classes, uint64_t cap, std::vector<uint64_t>* classBases)
{
    std::vector<uint32_t> idx(records.size());
    for (uint32_t i = 0; i < idx.size(); ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return records[a] > records[b]; });
    std::vector<uint32_t> owner(records.size(), classes - 1u);
    uint64_t sum = 0; uint32_t cls = 0;
    for (uint32_t i : idx) {
        if (cls + 1u >= classes) break;
        const uint64_t add = bases[i];
        if (sum != 0 && sum + add > cap) { ++cls; sum = 0; if (cls + 1u >= classes) break; }
        owner[i] = cls; sum += add;
    }
    if (classBases) { classBases->assign(classes, 0); for (size_t i = 0; i < owner.size(); ++i) (*classBases)[owner[i]] += bases[i]; }
    return owner;
}

void Context::packFiles(const std::vector<std::string>& inPrefixes, const std::vector<std::string>& outPrefixes, int verbose, bool hold)
{
    const double tStart = nowMs();
    const size_t nLibs = inPrefixes.size();
    if (nLibs == 0 || outPrefixes.size() != nLibs) throw std::runtime_error("pack: input/output prefix lists do not match");
    const uint32_t world = cfg.world_size ? cfg.world_size : 1, rank = cfg.rank;
    struct Lib {
        BinFile bf; ArchiveWriter aw;
        Batch b0; std::vector<uint8_t> block0; std::thread t0; std::string t0err; double t0ms = 0;
        bool haveBlock0 = false, block0Written = true, finished = false;
        bool block0InFile = false;               // its own thread has put block 0 at the head of the (still empty) archive
        std::atomic<bool> t0done{false};        // the block-0 thread has finished (its block can be written without waiting)
        struct Pending { std::vector<uint8_t> data; std::vector<uint64_t> sizes; std::vector<uint32_t> sigs; };
        std::vector<Pending> pending;
        std::mutex awMx;                        // block 0's thread decides under it whether it may head the archive itself; the main thread queues blocks under it
    };
    std::vector<std::unique_ptr<Lib>> libs;
    archives.clear(); archives.resize(nLibs);
    struct Work { uint32_t lib, sig; };
    std::vector<Work> work;
    for (size_t l = 0; l < nLibs; ++l) libs.emplace_back(new Lib());
    parallelFor((uint32_t)nLibs, std::min<uint32_t>((uint32_t)nLibs, hostThreads), [&](uint32_t l, uint32_t) {      // footers are parsed side by side
        Lib& L = *libs[l];
        L.bf.open(inPrefixes[l], par.minBinSize);
        archives[l].cfg = L.bf.config(); archives[l].head = L.bf.headData(); archives[l].qvz = L.bf.qvz();
        if (hold) L.aw.startInMemory(archives[l].cfg);
        else L.aw.start(world > 1 ? outPrefixes[l] + ".part" + std::to_string(rank) : outPrefixes[l], archives[l].cfg);
    });
    // Every bin is dealt up front (shardOwners: longest stream first, then the sums).  Round 3 had a work-stealing tail behind it -- the lightest
    // 15 % of the records in chunks that the ranks of a node claimed from one counter in /dev/shm as they ran out of work --, built, tested
    // and never the default: every chunk claimed behind a rank's first batch is a device batch of its own (150 ms and more for a few hundred
    // light bins), the shares dealt up front are within a few per cent of each other, and the two-rank rehearsal was no faster with it
    // (1 659 against 1 648 ms a step, profiles/r03_rehearse_2ranks_last_tail_*.json).  It left the tree in round 5.
    for (size_t l = 0; l < nLibs; ++l) {
        const auto& stdSigs = libs[l]->bf.stdSignatures();
        std::vector<uint64_t> w(stdSigs.size());
        for (uint32_t i = 0; i < stdSigs.size(); ++i) w[i] = libs[l]->bf.bins().at(stdSigs[i]).totalRecordsCount;
        std::vector<uint32_t> owner;
        if (splitRole != 0) {
            // weight classes (splitClasses): the longest streams of the job start at once and run beside everything else, instead of in front of it
            std::vector<uint64_t> bs(stdSigs.size());
            for (uint32_t i = 0; i < stdSigs.size(); ++i) bs[i] = libs[l]->bf.bins().at(stdSigs[i]).totalRawDnaSize;
            owner = splitClasses(w, bs, world, cfg.batch_bases ? cfg.batch_bases : (3072ull << 20));
        } else owner = shardOwners(w, world);
        for (uint32_t i = 0; i < stdSigs.size(); ++i) if (owner[i] == rank) work.push_back(Work{(uint32_t)l, stdSigs[i]});
        if (getenv("FS_TRACE")) fprintf(stderr, "[trace] library %zu: rank %u of %u (split role %u) packs %zu of %zu standard bins\n", l, rank, world, splitRole, work.size(), stdSigs.size());
    }
    haveArchive = true;
    // block 0 of every library (rank 0): merged small bins + N bin, compressed on host cores.  Its threads start when the
    // host tasks of the last batch are done: until then every core is needed by the front end, afterwards the host only
    // waits for the device.  When they have unpacked their bins the inputs are not needed any more and are unmapped by a
    // thread of their own, also inside the device's tail.
    // (With several device batches the threads start at once instead: the cores idle between batches anyway, and the
    // blocks of the finished batches can only be written -- and their memory released -- behind block 0.)
    const uint64_t budget = cfg.batch_bases ? cfg.batch_bases : (3072ull << 20);
    uint64_t stdBases = 0;
    for (const Work& w : work) stdBases += libs[w.lib]->bf.bins().at(w.sig).totalRawDnaSize;
    // (Round 3: with eight cores or more the block-0 threads start at once all the same -- a paired-end library's block 0 is
    // 0.8 s of one serial PPMd stream per 6 M pairs; behind a 1.3 s front end it WAS the end of the step (2.57 s), beside it
    // it costs the front end a sixteenth of its cores for that time.)
    const bool block0Early = hostCores >= 8;
    std::mutex gateMx; std::condition_variable gateCv; bool hostTasksDone = work.empty() || stdBases > budget || block0Early; size_t block0Unpacked = 0, block0Threads = 0;
    std::thread closer;
    if (rank == 0) {
        for (size_t l = 0; l < nLibs; ++l) {
            Lib& L = *libs[l];
            if (L.bf.smallSignatures().empty() && !L.bf.hasNBin()) continue;
            L.haveBlock0 = true; L.block0Written = false;
            const ArchiveParams* ap = &archives[l]; Lib* lp = &L;
            // unpack + compress in the background thread (BinFile::unpack only reads the mapped files)
            ++block0Threads;
            L.t0 = std::thread([this, lp, ap, &gateMx, &gateCv, &hostTasksDone, &block0Unpacked]() {
                { std::unique_lock<std::mutex> lk(gateMx); gateCv.wait(lk, [&]() { return hostTasksDone; }); }
                const double a = nowMs();
                try {
                    uint64_t rawDna = 0;
                    for (uint32_t sig : lp->bf.smallSignatures()) { lp->bf.unpack(sig, lp->b0, lp->b0.bins.empty()); rawDna += lp->bf.bins().at(sig).totalRawDnaSize; }
                    if (lp->bf.hasNBin()) { lp->bf.unpack(lp->bf.nSignature(), lp->b0, lp->b0.bins.empty()); rawDna += lp->bf.bins().at(lp->bf.nSignature()).totalRawDnaSize; }
                    lp->b0.bins[0].signature = lp->bf.nSignature(); lp->b0.bins[0].rawDnaSize = rawDna;
                    { std::lock_guard<std::mutex> lk(gateMx); ++block0Unpacked; }
                    gateCv.notify_all();
                    compressRawBlock(lp->b0, *ap, lp->block0);
                    // Block 0 heads the archive and nothing is written in front of it: this thread puts it there itself, while
                    // the device still walks the long streams (the writer belongs to this thread until it is joined).
                    { std::lock_guard<std::mutex> lk(lp->awMx); if (lp->aw.dataBytes() == 0 && lp->pending.empty()) { lp->aw.writeBlock(lp->block0.data(), lp->block0.size(), lp->bf.nSignature()); lp->block0InFile = true; } }
                } catch (const std::exception& e) {
                    lp->t0err = e.what();
                    { std::lock_guard<std::mutex> lk(gateMx); ++block0Unpacked; }       // never leave the closer waiting
                    gateCv.notify_all();
                }
                lp->t0ms = nowMs() - a;
                lp->t0done.store(true);
            });
        }
    }
    auto flush = [&](Lib& L, bool wait) {
        if (!L.block0Written) {
            if (!wait && !L.t0done.load()) return;
            L.t0.join(); if (!L.t0err.empty()) throw std::runtime_error(L.t0err);
            stats.block0_ms = std::max(stats.block0_ms, L.t0ms); stats.block0_bytes += L.block0.size(); stats.block0_records += L.b0.recs.size();
            if (!L.block0InFile) L.aw.writeBlock(L.block0.data(), L.block0.size(), L.bf.nSignature());
            L.block0Written = true; L.b0.clear(); L.block0.clear(); L.block0.shrink_to_fit();
        }
        for (auto& p : L.pending) { uint64_t off = 0; for (size_t k = 0; k < p.sizes.size(); ++k) { L.aw.writeBlock(p.data.data() + off, p.sizes[k], p.sigs[k]); off += p.sizes[k]; } }
        L.pending.clear();
    };
    Batch& batch = workBatch; std::vector<uint32_t> binArch; size_t next = 0, done = 0;
    try {
        while (next < work.size()) {
            batch.clear(); binArch.clear();
            double tio = nowMs();
            // choose the bins of this batch by their (known) unpacked size, unpack them in parallel, then concatenate
            const size_t first = next; uint64_t bases = 0;
            while (next < work.size()) {
                const uint64_t add = libs[work[next].lib]->bf.bins().at(work[next].sig).totalRawDnaSize;
                if (next > first && bases + add > budget) break;
                bases += add; ++next;
            }
            const uint32_t nb = (uint32_t)(next - first);
            ++stats.device_batches;
            // record arrays are placed: their sizes are in the .bmeta footer, so every bin knows its offsets up front and
            // is unpacked by the same host task that runs its front end (no barrier between the two stages)
            std::vector<uint64_t> seqBase(nb + 1, 0), headBase(nb + 1, 0), recBase(nb + 1, 0), quaBase(nb + 1, 0), hpBase(nb + 1, 0), dpBase(nb + 1, 0), weight(nb);
            // device-side unpack of the bases for the window search: the bins' .bdna bytes are kept as stored (FS_DEVICE_UNPACK=0: the ASCII bases go up)
            const bool packedD = deviceMatcher && deviceUnpack();
            // device-side read-id tokeniser: the headers stay packed too (FS_DEVICE_IDS=0: host tokeniser)
            bool packedH = !(getenv("FS_DEVICE_IDS") && atoi(getenv("FS_DEVICE_IDS")) == 0);
            for (uint32_t k = 0; k < nb; ++k) if (!libs[work[first + k].lib]->bf.usesHeaders()) packedH = false;
            // device-side quality path: lossless archives keep their scores packed (fs_gather_quality unpacks, orients and
            // orders them on the device); FS_DEVICE_QUALITY=0 keeps the host symbolisation (A/B runs)
            bool packedQ = !(getenv("FS_DEVICE_QUALITY") && atoi(getenv("FS_DEVICE_QUALITY")) == 0);
            // (all libraries of the batch must share one mode: lossless bytes, the (symbol, context) pairs of 8-bin / binary, or the
            // (context, state) words of --lossy archives -- those may each bring a codebook of their own)
            for (uint32_t k = 0; k < nb; ++k) { const uint32_t qmk = archives[work[first + k].lib].cfg.quaParams.method; if (qmk != archives[work[first].lib].cfg.quaParams.method) packedQ = false; }
            for (uint32_t k = 0; k < nb; ++k) {
                const BinInfo& bi = libs[work[first + k].lib]->bf.bins().at(work[first + k].sig);
                seqBase[k + 1] = seqBase[k] + bi.totalRawDnaSize; headBase[k + 1] = headBase[k] + bi.totalRawHeadSize; recBase[k + 1] = recBase[k] + bi.totalRecordsCount;
                quaBase[k + 1] = quaBase[k] + ((bi.totalQuaSize + 15u) & ~15ull);
                hpBase[k + 1] = hpBase[k] + ((bi.totalHeadSize + 15u) & ~15ull);
                dpBase[k + 1] = dpBase[k] + ((bi.totalDnaSize + 16u + 15u) & ~15ull);
                weight[k] = bi.totalRecordsCount;
            }
            // what a bin brings into a lane's staging buffer, roughly: its quality scores (packed, or a byte / a pair each), its
            // read ids, the descriptors and the small streams -- a fresh context sizes its lanes' pinned buffers by it up front
            binBases.assign(nb, 0);
            for (uint32_t k = 0; k < nb; ++k) binBases[k] = libs[work[first + k].lib]->bf.bins().at(work[first + k].sig).totalRawDnaSize;
            stageEstimate.assign(nb, 0); std::vector<uint64_t> archEstimate(nb, 0);
            for (uint32_t k = 0; k < nb; ++k) {
                const BinInfo& bi = libs[work[first + k].lib]->bf.bins().at(work[first + k].sig);
                const uint32_t qmk = archives[work[first + k].lib].cfg.quaParams.method;
                const uint64_t q = packedQ ? bi.totalQuaSize + (qmk == MET_QVZ ? bi.totalRawDnaSize / 4u * 4u + (2ull << 20) : 0ull) : (qmk == MET_NONE ? bi.totalRawDnaSize : (qmk == MET_QVZ ? 4ull : 2ull) * bi.totalRawDnaSize);
                stageEstimate[k] = q + (packedH ? bi.totalHeadSize : 2ull * bi.totalRawHeadSize) + 80ull * bi.totalRecordsCount;
                // (what the bin's block will roughly weigh: a quarter of what is staged -- PPMd on quality scores --, but QVZ's scores go up as a
                // word each and come back as a fraction of a bit: the estimate made a --lossy pack reserve 1.5 GB for a 0.4 GB archive and
                // pay 140 ms a step for making and dropping the pages)
                archEstimate[k] = (qmk == MET_QVZ ? bi.totalRawDnaSize / 16u : q / 4u) + ((packedH ? bi.totalHeadSize : 2ull * bi.totalRawHeadSize) + 80ull * bi.totalRecordsCount) / 4u;
            }
            if (seqBase[nb] > 0xFFFFFFF0ull || headBase[nb] > 0xFFFFFFF0ull || recBase[nb] > 0xFFFFFFF0ull) throw std::runtime_error("batch exceeds 4 GiB");
            batch.seq.resize(seqBase[nb]); batch.recs.resize(recBase[nb]);
            if (packedH) { batch.head.clear(); batch.headPacked.resize(hpBase[nb]); batch.headBit.resize(recBase[nb]); }
            else { batch.head.resize(headBase[nb]); batch.headPacked.clear(); batch.headBit.clear(); }
            {   // the largest bin's matcher tables: a read and at most one root copy per record, a construction per sub-tree
                // (fewer than records) + the pieces of the top-level one, a full window of warm-up entries per piece
                uint64_t mr = 0, ms = 0;
                for (uint32_t k = 0; k < nb; ++k) { mr = std::max<uint64_t>(mr, recBase[k + 1] - recBase[k]); ms = std::max<uint64_t>(ms, seqBase[k + 1] - seqBase[k]); }
                matchReserve.reads = 2 * mr + 16; matchReserve.seqBytes = ms; matchReserve.calls = mr + mr / 512 + 16;
                matchReserve.warm = (mr / 512 + 2) * (uint64_t)std::max(1u, par.maxLzWindowSize);
            }
            if (packedQ) { batch.qua.clear(); batch.quaPacked.resize(quaBase[nb]); batch.quaBit.resize(recBase[nb]); }
            else { batch.qua.resize(seqBase[nb]); batch.quaPacked.clear(); batch.quaBit.clear(); }
            if (packedD) { batch.dnaPacked.resize(dpBase[nb]); batch.dnaBit.resize(recBase[nb]); batch.dnaInfo.resize(recBase[nb]); }
            else { batch.dnaPacked.clear(); batch.dnaBit.clear(); batch.dnaInfo.clear(); }
            if (getenv("FS_TRACE")) fprintf(stderr, "[trace] batch of %u bins: record arrays placed at %.1f ms of the pack\n", nb, nowMs() - tStart);
            std::vector<Batch> graph(nb);                              // per bin: its stored graph (node indices local to it)
            for (size_t k = first; k < next; ++k) binArch.push_back(work[k].lib);
            stats.io_ms += nowMs() - tio;
            const bool lastBatchNow = next >= work.size();
            // what the archives will hold at the end, roughly: what they hold, what is pending, block 0 and a quarter of this
            // batch's staged bytes (PPMd on quality scores) -- their page-cache pages are made while the device works
            std::vector<uint64_t> aheadBytes(nLibs, 0);
            if (lastBatchNow) {
                for (uint32_t k = 0; k < nb; ++k) aheadBytes[work[first + k].lib] += archEstimate[k];
                for (size_t l = 0; l < nLibs; ++l) {
                    Lib& L = *libs[l];
                    std::lock_guard<std::mutex> lk(L.awMx);                      // (block 0's thread may be putting its block at the archive's head right now)
                    uint64_t have = L.aw.dataBytes() + (24u << 20);             // (+ block 0 and slack; what does not fit is written behind the extent)
                    for (const Lib::Pending& pd : L.pending) have += pd.data.size();
                    aheadBytes[l] += have;
                }
            }
            onHostTasksDone = [&, lastBatchNow]() {
                if (!lastBatchNow) return;
                for (size_t l = 0; l < nLibs; ++l) libs[l]->aw.reserveAhead(getenv("FS_AHEAD_BYTES") ? (uint64_t)atoll(getenv("FS_AHEAD_BYTES")) : aheadBytes[l]);      // (the variable: tests of a short estimate)
                { std::lock_guard<std::mutex> lk(gateMx); hostTasksDone = true; }
                gateCv.notify_all();
                closer = std::thread([&]() {                       // the mapped inputs have been read for the last time
                    { std::unique_lock<std::mutex> lk(gateMx); gateCv.wait(lk, [&]() { return block0Unpacked >= block0Threads; }); }
                    parallelFor((uint32_t)nLibs, std::min<uint32_t>((uint32_t)nLibs, 8u), [&](uint32_t l, uint32_t) { libs[l]->bf.close(); });
                });
            };
            releaseEarly = cfg.one_shot != 0 && lastBatchNow;
            // (... and what only the host tasks and the staging used: the bins' streams, the encoders' work buffers)
            onAllStaged = [&]() { const double tr = nowMs(); batch.release(); std::vector<Batch>().swap(graph); std::vector<BinStreams>().swap(streamPool); encoders.clear(); if (getenv("FS_TRACE")) fprintf(stderr, "[trace] the batch's record arrays given back in %.1f ms, at %.1f ms of the pack\n", nowMs() - tr, nowMs() - tStart); };
            compressBins(nb, weight, binArch, [&](uint32_t k, BinEncoder& enc, BinStreams& out, BinIn& info, uint64_t& recBytes) {
                const Work& w = work[first + k];
                const double tu = getenv("FS_BIN_TRACE") ? nowMs() : 0;
                if (tu > 0 && k < 3) fprintf(stderr, "[bin] task %u starts at %.1f ms of the pack\n", k, tu - tStart);
                libs[w.lib]->bf.unpackPlaced(w.sig, batch, seqBase[k], headBase[k], (uint32_t)recBase[k], graph[k], packedQ ? (int64_t)quaBase[k] : -1, packedH ? (int64_t)hpBase[k] : -1, packedD ? (int64_t)dpBase[k] : -1);
                if (tu > 0 && recBase[k + 1] - recBase[k] >= 40000) fprintf(stderr, "[bin] %llu records: unpack %.1f ms\n", (unsigned long long)(recBase[k + 1] - recBase[k]), nowMs() - tu);
                info = graph[k].bins.at(0);
                enc.encodeLz(batch, graph[k], info, archives[w.lib], out);
                if (packedQ) { out.quaPacked = batch.quaPacked.data() + quaBase[k]; out.quaPackedBytes = libs[w.lib]->bf.bins().at(w.sig).totalQuaSize; }
                if (packedH) { out.headPacked = batch.headPacked.data() + hpBase[k]; out.headPackedBytes = libs[w.lib]->bf.bins().at(w.sig).totalHeadSize; }
                recBytes = 2ull * (seqBase[k + 1] - seqBase[k]) + (headBase[k + 1] - headBase[k]);
                graph[k] = Batch();
            });
            const double tRoute = nowMs();
            const bool lastBatch = lastBatchNow;
            if (!lastBatch) {
                for (size_t b = 0; b < nb;) {                          // route the blocks to their libraries (runs of equal lib)
                    const uint32_t l = binArch[b]; Lib::Pending p;
                    size_t e = b; uint64_t bytes = 0;
                    while (e < nb && binArch[e] == l) { p.sizes.push_back(blockSizes[e]); p.sigs.push_back(binInfo[e].signature); bytes += blockSizes[e]; ++e; }
                    p.data.resize(bytes);
                    uint64_t off = 0;
                    for (size_t k = b; k < e; ++k) { memcpy(p.data.data() + off, blockData((uint32_t)k), blockSizes[k]); off += blockSizes[k]; }
                    { std::lock_guard<std::mutex> lk(libs[l]->awMx); libs[l]->pending.push_back(std::move(p)); } b = e;
                }
                tio = nowMs();
                for (auto& L : libs) flush(*L, false);
                stats.io_ms += nowMs() - tio;
            } else {
                // last batch: every library is finished by a task of its own -- block 0, what is pending, then this batch's
                // blocks straight from the slice buffers (no routing copy), then the .cmeta footer
                tio = nowMs();
                std::vector<std::pair<size_t, size_t>> runOf(nLibs, {0, 0});
                for (size_t b2 = 0; b2 < nb;) { size_t e = b2; while (e < nb && binArch[e] == binArch[b2]) ++e; runOf[binArch[b2]] = {b2, e}; b2 = e; }
                std::mutex statMx;
                parallelFor((uint32_t)nLibs, std::min<uint32_t>((uint32_t)nLibs, hostThreads), [&](uint32_t l, uint32_t) {
                    Lib& L = *libs[l];
                    if (!L.block0Written) {
                        L.t0.join(); if (!L.t0err.empty()) throw std::runtime_error(L.t0err);
                        { std::lock_guard<std::mutex> g(statMx); stats.block0_ms = std::max(stats.block0_ms, L.t0ms); stats.block0_bytes += L.block0.size(); stats.block0_records += L.b0.recs.size(); }
                        if (!L.block0InFile) L.aw.writeBlock(L.block0.data(), L.block0.size(), L.bf.nSignature());
                        L.block0Written = true; L.b0.clear(); L.block0.clear(); L.block0.shrink_to_fit();
                    }
                    for (auto& p : L.pending) { uint64_t off = 0; for (size_t k = 0; k < p.sizes.size(); ++k) { L.aw.writeBlock(p.data.data() + off, p.sizes[k], p.sigs[k]); off += p.sizes[k]; } }
                    L.pending.clear();
                    {
                        std::vector<const uint8_t*> ptrs; std::vector<uint64_t> szs; std::vector<uint32_t> sgs;
                        for (size_t k = runOf[l].first; k < runOf[l].second; ++k) { ptrs.push_back(blockData((uint32_t)k)); szs.push_back(blockSizes[k]); sgs.push_back(binInfo[k].signature); }
                        L.aw.writeBlocks(ptrs, szs, sgs, std::max(1u, hostThreads / (uint32_t)std::max<size_t>(1, nLibs)));
                    }
                    L.aw.finish(archives[l].head, archives[l].qvz);
                    L.finished = true;
                });
                stats.io_ms += nowMs() - tio;
            }
            done += nb;
            if (getenv("FS_TRACE")) fprintf(stderr, "[trace] route+write %.1f ms\n", nowMs() - tRoute);
            if (verbose) { fprintf(stderr, "\rParts processed: %zu (%zu%%) ", done, work.empty() ? 100 : done * 100 / work.size()); fflush(stderr); }
        }
        const double tio = nowMs();
        if (getenv("FS_TRACE")) fprintf(stderr, "[trace] before final flush at %.1f ms\n", nowMs() - tStart);
        for (size_t l = 0; l < nLibs; ++l) {
            if (!libs[l]->finished) { flush(*libs[l], true); libs[l]->aw.finish(archives[l].head, archives[l].qvz); }      // libraries without standard bins on this rank
            stats.cdata_bytes += libs[l]->aw.dataBytes();
        }
        if (verbose) fprintf(stderr, "\n");
        if (verbose == 1) for (size_t l = 0; l < nLibs; ++l) libs[l]->aw.printStreamSizes(stdout);
        stats.io_ms += nowMs() - tio;
    } catch (...) {
        onHostTasksDone = nullptr; onAllStaged = nullptr; releaseEarly = false;
        { std::lock_guard<std::mutex> lk(gateMx); hostTasksDone = true; }
        gateCv.notify_all();
        for (auto& L : libs) if (L->t0.joinable()) L->t0.join();
        if (closer.joinable()) closer.join();
        throw;
    }
    onHostTasksDone = nullptr; onAllStaged = nullptr; releaseEarly = false;
    { std::lock_guard<std::mutex> lk(gateMx); hostTasksDone = true; }          // no standard bins at all: block 0 starts here
    gateCv.notify_all();
    if (closer.joinable()) closer.join();
    if (hold) {       // the blocks stay with the context; every archive's block table in its final order: block 0, then ascending signature
        shards.clear(); shards.resize(nLibs);
        for (size_t l = 0; l < nLibs; ++l) {
            Lib& L = *libs[l]; Shard& sh = shards[l];
            if (!L.bf.smallSignatures().empty() || L.bf.hasNBin()) sh.order.push_back(L.bf.nSignature());
            for (uint32_t sg : L.bf.stdSignatures()) sh.order.push_back(sg);
            sh.arch = archives[l];
            sh.aw.reset(new ArchiveWriter(std::move(L.aw)));
            sh.have = true;
        }
    }
    {   // whatever is still mapped (no standard bins on this rank): one task per library
        const double tc = nowMs();
        parallelFor((uint32_t)nLibs, std::min<uint32_t>((uint32_t)nLibs, hostThreads), [&](uint32_t l, uint32_t) { libs[l].reset(); });
        if (getenv("FS_TRACE")) fprintf(stderr, "[trace] close inputs %.1f ms\n", nowMs() - tc);
    }
    stats.total_ms += nowMs() - tStart;
    if (getenv("FS_TRACE")) fprintf(stderr, "[trace] packFiles total %.1f ms\n", nowMs() - tStart);
}

// ------------------------------------------------------------------------------------------------
void Context::shardPack(const std::vector<std::string>& inPrefixes)
{
    shards.clear();
    packFiles(inPrefixes, std::vector<std::string>(inPrefixes.size(), std::string("(held)")), 0, true);
}

void Context::shardTable(size_t lib, std::vector<uint32_t>& sigs, std::vector<uint64_t>& sizes) const
{
    if (lib >= shards.size() || !shards[lib].have) throw std::runtime_error("no held pack: call the shard pack first");
    const Shard& shard = shards[lib];
    sigs = shard.order; sizes.assign(sigs.size(), 0);
    std::map<uint32_t, size_t> place;
    for (size_t i = 0; i < sigs.size(); ++i) place[sigs[i]] = i;
    for (const ArchiveWriter::HeldBlock& b : shard.aw->heldBlocks()) {
        const auto it = place.find(b.signature);
        if (it == place.end()) throw std::runtime_error("held block without a place in the archive");
        sizes[it->second] = b.size;
    }
}

void Context::shardWrite(size_t lib, const std::string& outPrefix, const std::vector<uint64_t>& allSizes)
{
    if (lib >= shards.size() || !shards[lib].have) throw std::runtime_error("no held pack: call the shard pack first");
    Shard& shard = shards[lib];
    std::vector<uint32_t> sigs; std::vector<uint64_t> own;
    shardTable(lib, sigs, own);
    if (allSizes.size() != sigs.size()) throw std::runtime_error("size table does not match the archive's block table");
    std::vector<uint64_t> off(sigs.size() + 1, 0);
    for (size_t i = 0; i < sigs.size(); ++i) {
        if (own[i] && own[i] != allSizes[i]) throw std::runtime_error("size table disagrees with the held blocks");
        // (a block is never empty: its header alone is 42 bytes -- a zero here is a bin that no rank packed)
        if (allSizes[i] == 0) throw std::runtime_error("size table: a block of the archive was packed by no rank");
        off[i + 1] = off[i] + allSizes[i];
    }
    // O_CREAT without O_TRUNC and positional writes: the ranks need no order among themselves (nobody cuts the file below
    // its final size; rank 0 cuts a longer file of an earlier run down to it)
    const std::string name = outPrefix + ".cdata";
    const int fd = ::open(name.c_str(), O_CREAT | O_RDWR, 0644);
    if (fd < 0) throw std::runtime_error("Cannot open file: " + name);
    bool ok = true;
    std::map<uint32_t, size_t> place;
    for (size_t i = 0; i < sigs.size(); ++i) place[sigs[i]] = i;
    for (const ArchiveWriter::HeldBlock& b : shard.aw->heldBlocks()) {
        const uint8_t* p = shard.aw->heldData() + b.offset; uint64_t left = b.size, at = off[place.at(b.signature)];
        while (ok && left) { const ssize_t w = ::pwrite(fd, p, left, (off_t)at); if (w <= 0) { ok = false; break; } p += w; left -= (uint64_t)w; at += (uint64_t)w; }
    }
    if (cfg.rank == 0) ok = ok && ::ftruncate(fd, (off_t)off[sigs.size()]) == 0;
    if (::close(fd) != 0 || !ok) throw std::runtime_error("Cannot write " + name);
    // (the held bytes were counted when they were packed: packFiles)
    if (cfg.rank == 0) shard.aw->writeMeta(outPrefix, allSizes, sigs, shard.arch.head, shard.arch.qvz);
    shard = Shard();
}

}  // namespace fs
