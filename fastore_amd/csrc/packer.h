// Host orchestration of the hot path: batches of bins -> front end (host threads) -> HIP engine ->
// blocks; plus the merged small-bins/N block and the archive writer (drop-in boundary).
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <functional>
#include <memory>
#include <string>
#include <future>
#include <chrono>
#include <thread>
#include <vector>
#include "../../include/fastore_amd.h"
#include "binfile.h"
#include "engine.h"
#include "format.h"
#include "frontend.h"

namespace fs {

// the reference's -v statistics (CompressorModule.cpp:357-387): compressed bytes per stream summed over the standard
// blocks (block header, FastqCompressor.cpp:1216-1219) and the four stream sizes of the raw block (:3592-3596), read
// from the blocks' own headers -- so they can be taken from a finished archive as well (merged multi-GPU parts)
class StreamSizeStats {
public:
    void start(const ArchiveTypeRaw& type, const MinimizerParametersRaw& mp);
    void addBlock(const uint8_t* header, uint64_t size, uint32_t signature);     // `header`: at least min(size, headerBytes()) bytes of the block
    uint64_t headerBytes() const { return 34 + (hasHeaders_ ? 8 : 0) + 16ull * nStreams_; }
    void print(FILE* to) const;
private:
    std::vector<uint64_t> streamComp_; uint64_t rawComp_[4] = {0, 0, 0, 0}; bool haveRaw_ = false;
    uint32_t nStreams_ = 0; bool hasHeaders_ = false; uint32_t rawSignature_ = 0;
};

// .cmeta/.cdata writer: ArchiveFileWriter (fastore_pack/ArchiveFile.cpp:21-204)
// Bins -> ranks for one library packed by several GPUs: longest-processing-time-first over the per-signature
// record totals of the .bmeta footer (fastore_bin/BinFile.h:57-79) -- heaviest bin to the least loaded rank, ties to
// the lower index / rank.  A pure function of the footer: every rank computes the same table, nothing is exchanged.
std::vector<uint32_t> shardOwners(const std::vector<uint64_t>& weights, uint32_t world);
// The weight classes of a library packed by several pipelines on ONE device (capi.cpp: packSplit): the heaviest bins, as many as one device
// batch holds (cap bases), are class 0, the next batch's worth class 1 (three classes), all the others the last class.  records / bases: per
// standard bin, in the order of the library's signature list; classBases (optional): the bases every class ends up with.
std::vector<uint32_t> splitClasses(const std::vector<uint64_t>& records, const std::vector<uint64_t>& bases, uint32_t classes, uint64_t cap, std::vector<uint64_t>* classBases = nullptr);
// Worker slots shared by the pipelines of ONE split pack (capi.cpp: packSplit): a bin's front end holds a slot; pipelines that wait are served
// in the order packSplit gives their classes, a class holding no more slots at once than its cap.  (Round 5: with a fixed share of the threads
// each, the heaviest class's 183 bins kept six threads busy for 9 s of a 25 M-pair pack while the lightest class's twelve were done after 2.7 s.)
struct HostGate {
    std::mutex mx; std::condition_variable cv; uint32_t free;
    uint32_t waiting[4] = {0, 0, 0, 0}, held[4] = {0, 0, 0, 0};
    uint32_t rank[4] = {0, 1, 2, 3};      // a class's place in the order the waiting pipelines are served (0 first)
    uint32_t cap[4] = {0, 0, 0, 0};       // the slots a class may hold at once (0: any number)
    explicit HostGate(uint32_t slots) : free(slots) {}
    bool eligible(uint32_t c) const { return cap[c] == 0 || held[c] < cap[c]; }
    void acquire(uint32_t cls)
    {
        std::unique_lock<std::mutex> lk(mx);
        ++waiting[cls];
        cv.wait(lk, [&]() {
            if (!free || !eligible(cls)) return false;
            for (uint32_t c = 0; c < 4; ++c) if (rank[c] < rank[cls] && waiting[c] && eligible(c)) return false;
            return true;
        });
        --waiting[cls]; --free; ++held[cls];
        if (free) cv.notify_all();
    }
    void release(uint32_t cls) { { std::lock_guard<std::mutex> g(mx); ++free; --held[cls]; } cv.notify_all(); }
};

class ArchiveWriter {
public:
    ArchiveWriter() = default;
    ArchiveWriter(ArchiveWriter&& o);
    ~ArchiveWriter();
    void start(const std::string& prefix, const BinModuleConfigRaw& cfg);
    // keep the blocks in memory instead (bin-sharded packs: the blocks' places in the archive depend on the other ranks' sizes)
    void startInMemory(const BinModuleConfigRaw& cfg);
    struct HeldBlock { uint32_t signature; uint64_t offset, size; };
    const std::vector<HeldBlock>& heldBlocks() const { return held_; }
    const uint8_t* heldData() const { return mem_.data(); }
    // .cmeta of an archive whose blocks (sizes, signatures in archive order) were written by several ranks
    void writeMeta(const std::string& prefix, const std::vector<uint64_t>& sizes, const std::vector<uint32_t>& sigs, const HeaderStats& head, const QvzModel& qvz);
    void writeBlock(const uint8_t* data, uint64_t size, uint32_t signature);
    // the same for a run of blocks that follow each other in the archive: their places are known, so several threads put
    // them there with positional writes (one buffered writer moves ~4 GB/s: 0.1 s of a 0.45 GB archive at the very end of a step)
    void writeBlocks(const std::vector<const uint8_t*>& data, const std::vector<uint64_t>& sizes, const std::vector<uint32_t>& signatures, uint32_t threads);
    void finish(const HeaderStats& head, const QvzModel& qvz);
    // The archive's first `bytes` bytes are given their page-cache pages NOW, by a thread of the writer's own (the host is
    // idle while the device walks the long streams; the final copy of the blocks then finds its pages mapped instead of
    // faulting 10^5 of them in at the very end of the step).  An estimate: finish() cuts the file to what was written.
    void reserveAhead(uint64_t bytes);
    uint64_t dataBytes() const { return dataBytes_; }
    void printStreamSizes(FILE* to) const { sizeStats_.print(to); }
private:
    StreamSizeStats sizeStats_;
    std::string prefix_;
    FILE *meta_ = nullptr, *data_ = nullptr;
    bool inMemory_ = false; std::vector<uint8_t> mem_; std::vector<HeldBlock> held_;
    ArchiveConfigRaw conf_{};
    std::vector<uint64_t> sizes_;
    std::vector<uint32_t> sigs_;
    uint64_t dataBytes_ = 0;
    uint8_t* pre_ = nullptr; size_t preLen_ = 0; std::thread preTh_;
    void dropAhead();
};

struct Context {
    fsgpu_config cfg{};
    PackParams par;
    fsengine::Device* dev = nullptr;
    // A one-shot context starts its device (HIP runtime, code objects, arena pool, first lane: 0.2 s, seconds on a box whose
    // driver is clearing memory) on a thread of its own while the front end of the first bins runs: device() waits for it
    // and throws what went wrong, deviceReadyWithin() asks.
    std::shared_future<std::string> devPending; std::atomic<bool> devAsync{false}; std::mutex devMx; std::string devError;
    fsengine::Device* device();
    bool deviceReadyWithin(int ms) { return !devAsync.load() || devPending.wait_for(std::chrono::milliseconds(ms)) == std::future_status::ready; }
    std::vector<ArchiveParams> archives;          // one per library being packed (index 0 for the single-library calls)
    bool haveArchive = false;
    std::string err;
    // result of compressBatch: block b of the batch is blockSizes[b] bytes at blockData(b); gatherBlocks() makes them contiguous
    std::vector<uint64_t> blockSizes;
    std::vector<uint32_t> blockSlice; std::vector<uint64_t> blockOff;
    std::vector<std::vector<uint8_t>> sliceBlocks;
    std::vector<uint8_t> blocks;
    const uint8_t* blockData(uint32_t b) const { return sliceBlocks[blockSlice[b]].data() + blockOff[b]; }
    void gatherBlocks();
    std::vector<fsengine::Device*> lanes;         // lanes[0] == dev; further engine instances for the pipelined batches
    uint64_t equalizeStage = 0; uint32_t equalizeLanes = 0;     // pending: make the lanes' buffers alike
    void equalizeNow();
    fsengine::Device* lane(uint32_t i);
    fsgpu_stats stats{};
    fsengine::BatchTiming timing;
    uint32_t hostThreads = 1;
    uint32_t hostCores = 0;                       // cores this process may really use (0: unknown); hostThreads is 1.5 x that by default
    // buffers kept across calls (their pages stay mapped: no first-touch faults or munmap churn per batch)
    Batch workBatch;
    std::vector<BinStreams> streamPool;

    // standard bins of `batch` -> blocks/blockSizes (bin order); binArch[b] = index into `archives`
    void compressBatch(const Batch& batch, const std::vector<uint32_t>& binArch);
    // the same with the bins produced on demand: produce(b, encoder, streams, info, recBytes) runs the front end of bin b
    // (after unpacking it, for the file path) on one of the host threads
    typedef std::function<void(uint32_t, BinEncoder&, BinStreams&, BinIn&, uint64_t&)> BinProducer;
    void compressBins(uint32_t nBins, const std::vector<uint64_t>& weight, const std::vector<uint32_t>& binArch, const BinProducer& produce);
    std::vector<uint64_t> binBases;               // optional, per bin of the coming compressBins call: its bases (unpacked), for the slices' address budget
    std::vector<uint64_t> stageEstimate;          // optional, per bin of the coming compressBins call: bytes it will bring into a lane's staging buffer
    std::vector<BinIn> binInfo;                   // per bin of the last compressBins call
    std::function<void()> onHostTasksDone;        // called by compressBins when its host tasks are done (the device may still run)
    // A context that packs ONCE (the CLI): what the last batch does not need any more is given back while the device still walks the long
    // streams -- the batch's record arrays when its last slice has been staged (onAllStaged, on a thread of its own), a lane's staging
    // buffer when its slice is done -- instead of at the process's end, where freeing 8 GB of pages was 0.5 s of its 2.4 s
    // (profiles/r05_cli_trace.txt).  Set by packFiles for the last batch.
    bool releaseEarly = false;
    std::function<void()> onAllStaged;
    std::vector<std::unique_ptr<BinEncoder>> encoders;   // one per host thread
    // device-side window search (matcher.hip): a matcher lane per host thread, made on first use; deviceMatcher off = host scan
    std::vector<fsengine::MatchLane*> matchLanes;
    bool deviceMatcher = true;
    // two pipelines on one device for libraries of several batches (capi.cpp: packSplit): 0 = none, 1 = this context packs the
    // heaviest bins (one batch's worth: their streams are the longest of the job), 2 = all the others
    uint32_t splitRole = 0;
    struct { size_t reads = 0, seqBytes = 0, calls = 0, warm = 0; } matchReserve;     // bounds of the largest bin of the coming batch (0: grow on demand)
    std::atomic<uint64_t> matchedReads{0}, matchUs{0}, matchKernelUs{0}, matchBasesUp{0}, matchUnpackedReads{0};
    HostGate* hostGate = nullptr; uint32_t gateClass = 0;      // a split pack: the pipelines' shared worker slots, this pipeline's class (0 = heaviest)
    uint32_t sliceThreads = 0;                                 // ... and the number of bins a first slice is made of (0: hostThreads)
    std::atomic<int> searchesInFlight{0};      // host threads inside a device window search right now (the gate of the lighter bins' searches)
    // the device unpacks the bases of the window search itself from the bin's .bdna bytes (FS_DEVICE_UNPACK=0: ASCII bases go up)
    static bool deviceUnpack() { const char* e = getenv("FS_DEVICE_UNPACK"); return !(e && atoi(e) == 0); }      // (read per batch)
    MatchFn matcherFor(uint32_t tid);
    MateFn mateMatcherFor(uint32_t tid);          // the mate searches of host thread `tid` (paired-end bins): the same lane
    std::atomic<uint64_t> matedPairs{0}, mateUs{0}, mateKernelUs{0};
    void matcherCheck(const std::string& inPrefix, uint64_t& reads, uint64_t& differing);
    // parity check of the device mate search: every standard bin of a paired-end library through the host search and fs_match_mates
    void mateMatcherCheck(const std::string& inPrefix, uint64_t& pairs, uint64_t& differing);
    // parity check of the device tokeniser: every standard bin's read ids through the host tokeniser and through fs_tokenise_ids
    struct MateDispatcher;
    void emitCheck(const std::string& inPrefix, uint64_t& ops, uint64_t& streamsCompared, uint64_t& differing);
    void tokeniserCheck(const std::string& inPrefix, uint64_t& ids, uint64_t& differingBins);
    // merged small bins + N bin (batch with ONE bin, records already in stored order): RawCompressorSE/PE
    void compressRawBlock(Batch& batch, const ArchiveParams& arch, std::vector<uint8_t>& out) const;
    // `fastore_pack e` for one or several libraries; bins of all libraries share the device batches
    // verbose: 0 quiet, 1 = the reference's -v (progress on stderr, StreamSizes on stdout), 2 = progress only
    void packFiles(const std::vector<std::string>& inPrefixes, const std::vector<std::string>& outPrefixes, int verbose, bool hold = false);
    // bin-sharded pack of ONE library, in three steps (the middle one is the caller's: an all-gather / all-reduce of the
    // size table over RCCL, or a sum over the contexts of one process): shardPack codes this rank's bins and holds the
    // blocks; shardTable lists every block of the archive in its final (-t1) order with this rank's sizes (0 elsewhere);
    // shardWrite puts the held blocks at their offsets in <out>.cdata (rank 0 also writes <out>.cmeta).
    // (a SET of libraries goes through the same three steps in one device pipeline: lib = index into the prefixes of shardPack)
    void shardPack(const std::vector<std::string>& inPrefixes);
    void shardTable(size_t lib, std::vector<uint32_t>& sigs, std::vector<uint64_t>& sizes) const;
    void shardWrite(size_t lib, const std::string& outPrefix, const std::vector<uint64_t>& allSizes);
    struct Shard { std::unique_ptr<ArchiveWriter> aw; std::vector<uint32_t> order; ArchiveParams arch; bool have = false; };
    std::vector<Shard> shards;
};

void parseHeaderFields(const uint8_t* p, size_t n, bool pairedEnd, HeaderStats& out);
void serializeHeaderFields(const HeaderStats& head, bool pairedEnd, std::vector<uint8_t>& out);

}  // namespace fs
