// On-disk structures of the bin/rebin -> pack hand-off (.bmeta/.bdna/.bqua/.bhead) and of the
// archive (.cmeta/.cdata), plus the in-memory batch layout handed to the hot path.
//
// The reference serialises several structs raw (x86-64 SysV layout); the mirrors below have the
// same field order and types so that sizeof/offsetof agree:
//   BinModuleConfig     /root/reference/fastore/fastore_bin/Params.h:167-193  (BinFile.cpp:323,684)
//   BlockMetaData       /root/reference/fastore/fastore_bin/BinFile.h:27-48   (11 x u64)
//   ArchiveConfig       /root/reference/fastore/fastore_pack/ArchiveFile.h:29-34 (ArchiveFile.cpp:123)
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <new>
#include <set>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace fs {

struct ArchiveTypeRaw { uint8_t readType, qualityOffset, readsHaveHeaders; };
struct CategorizerParametersRaw { uint32_t minBlockBinSize; };
struct MinimizerParametersRaw { uint8_t signatureLen, skipZoneLen, signatureMaskCutoffBits; char dnaSymbolOrder[5]; };
struct MinimizerFilteringParametersRaw { uint8_t filterLowQualitySignatures, lowQualityThreshold; };
struct QvOptionsRaw { uint8_t verbose, stats, uncompressed, distortion; const char* dist_file; const char* uncompressed_name; double D; };
struct QualityCompressionParamsRaw { uint8_t method, binaryThreshold; QvOptionsRaw qvzOpts; };
struct HeadersCompressionParamsRaw { uint8_t preserveComments; };
struct BinModuleConfigRaw {
    ArchiveTypeRaw archiveType;
    CategorizerParametersRaw catParams;
    MinimizerParametersRaw minimizer;
    MinimizerFilteringParametersRaw minFilter;
    QualityCompressionParamsRaw quaParams;
    HeadersCompressionParamsRaw headParams;
    uint64_t fastqBlockSize;
    uint32_t binningLevel;
    uint8_t binningType;
};
struct ArchiveConfigRaw { ArchiveTypeRaw archType; MinimizerParametersRaw minParams; QualityCompressionParamsRaw quaParams; };
static_assert(sizeof(QualityCompressionParamsRaw) == 40, "layout");
static_assert(sizeof(BinModuleConfigRaw) == 88, "layout");
static_assert(sizeof(ArchiveConfigRaw) == 56, "layout");

struct BlockMetaDataRaw {           // BinaryBinDescriptor + file offsets
    uint64_t metaSize, dnaSize, quaSize, headSize, recordsCount, rawDnaSize, rawHeadSize;
    uint64_t metaFileOffset, dnaFileOffset, quaFileOffset, headFileOffset;
};
static_assert(sizeof(BlockMetaDataRaw) == 88, "layout");

enum QualityMethod { MET_NONE = 0, MET_BINARY = 1, MET_8BIN = 2, MET_QVZ = 3 };
enum ReadType { READ_SE = 0, READ_PE = 1 };

// read-id field statistics produced by fastore_bin (fastore_bin/Stats.h:40-71)
struct HeaderField {
    bool isConst = false, isNumeric = false;
    char separator = 0;
    uint64_t minValue = (uint64_t)-1, maxValue = 0;
    std::vector<std::string> possibleValues;    // kept in std::set order (sorted, unique)
};
struct HeaderStats { std::vector<HeaderField> fields; uint32_t pairedEndFieldIdx = 0; };

struct BinInfo {
    std::vector<BlockMetaDataRaw> blocks;
    uint64_t totalMetaSize = 0, totalDnaSize = 0, totalQuaSize = 0, totalHeadSize = 0;
    uint64_t totalRawDnaSize = 0, totalRawHeadSize = 0, totalRecordsCount = 0;
};

// ---------------------------------------------------------------------------------------------
// Unpacked records of a batch of bins: structure-of-arrays, the layout that is copied to HBM.
// Bases and qualities share offsets (seqOff); a PE record stores mate 1 then mate 2 contiguously.
struct Rec {
    uint32_t seqOff;      // into Batch::seq / Batch::qua
    uint32_t headOff;     // into Batch::head
    uint16_t seqLen, auxLen, minimPos;
    uint8_t headLen, flags;   // flags: bit0 = read is reverse-complemented, bit1 = pair swapped
};
enum { FLAG_REVERSE = 1, FLAG_SWAPPED = 2 };

struct NodeIn {           // a node of the stored match graph (fastore_rebin/NodesPacker.cpp:567-679)
    uint32_t rec;
    uint32_t emBegin, emCount;       // exact-match group: records emRecs[emBegin .. +emCount)
    uint32_t treeBegin, treeCount;   // sub-tree groups: trees[treeBegin .. +treeCount)
};
struct TreeIn { uint32_t signatureId; int32_t mainSignaturePos; uint32_t nodeBegin, nodeCount; };

struct BinIn {            // one bin = one future archive block
    uint32_t signature;
    uint32_t minLen, maxLen;
    uint64_t rawDnaSize;
    uint32_t recBegin, recCount;     // records of the bin in Batch::recs
    uint32_t topBegin, topCount;     // top-level nodes: Batch::topNodes[topBegin .. +topCount) -> Batch::nodes
    uint64_t dnaPackedOff, dnaPackedBytes;   // the bin's .bdna bytes in Batch::dnaPacked (0 bytes: not kept)
};

void fs_advise_huge(void* p, size_t bytes);      // madvise(MADV_HUGEPAGE) where the platform has it (binfile.cpp)

// allocator whose resize() leaves new elements uninitialised: the big record arrays are written exactly once
// (large arrays come 2 MiB-aligned and marked for transparent huge pages: the record arrays of a 10 M-read library are
// gigabytes that a fresh process touches once -- 512 times fewer page faults at its start, and pages to hand back at its end)
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { typedef NoInitAlloc<U> other; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
    T* allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T), huge = (size_t)2 << 20;
        if (bytes < 4 * huge) { void* p = malloc(bytes ? bytes : 1); if (!p) throw std::bad_alloc(); return (T*)p; }
        void* p = aligned_alloc(huge, (bytes + huge - 1) & ~(huge - 1));
        if (!p) throw std::bad_alloc();
        fs_advise_huge(p, (bytes + huge - 1) & ~(huge - 1));
        return (T*)p;
    }
    void deallocate(T* p, size_t) noexcept { free(p); }
    template <class U> void construct(U* p) noexcept { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
typedef std::vector<uint8_t, NoInitAlloc<uint8_t>> ByteVec;

struct Batch {
    ByteVec seq, qua, head;
    std::vector<Rec, NoInitAlloc<Rec>> recs;
    // Device-side quality path (lossless archives packed from .b* files): the quality strings stay as they are stored,
    // six bits per score, MSB first (fastore_bin/FastqPacker.cpp:157-287): quaPacked = the bins' .bqua bytes back to back,
    // quaBit[r] = bit offset of record r's first score from the start of ITS BIN's bytes (a PE record's second mate
    // follows the first).  `qua` is left empty then: the scores are unpacked, oriented and put in emission order by
    // the fs_gather_quality kernel.
    ByteVec quaPacked;
    std::vector<uint32_t, NoInitAlloc<uint32_t>> quaBit;
    // Device-side read-id tokeniser: the headers stay as stored too (.bhead: a length byte, then 7 bits per character behind
    // an implied '@', fastore_bin/FastqPacker.cpp:157-287): headPacked = the bins' .bhead bytes, headBit[r] = bit offset of
    // record r's first stored character in its bin's bytes (Rec::headLen is set, `head` stays empty)
    ByteVec headPacked;
    std::vector<uint32_t, NoInitAlloc<uint32_t>> headBit;
    // Device-side unpack of the bases for the window search (matcher.hip: fs_unpack_planes): the bins' .bdna bytes as stored,
    // dnaBit[r] = bit offset of record r's first stored base in ITS BIN's bytes, dnaInfo[r] = the rest of fsdev::PackedRead
    // (an exact-match record names its main record's bases: it has none of its own).  `seq` is filled as ever: the host's
    // tree building reads it.
    ByteVec dnaPacked;
    std::vector<uint32_t, NoInitAlloc<uint32_t>> dnaBit, dnaInfo;
    std::vector<NodeIn> nodes;
    std::vector<uint32_t> topNodes;
    std::vector<uint32_t> emRecs;
    std::vector<TreeIn> trees;
    std::vector<BinIn> bins;
    // append `o` (whole bins) behind this batch, re-basing every index
    void append(const Batch& o);
    // give the memory back (clear() keeps the capacity for the next batch)
    void release() { Batch empty; std::swap(*this, empty); }
    void clear() { seq.clear(); qua.clear(); head.clear(); recs.clear(); quaPacked.clear(); quaBit.clear(); headPacked.clear(); headBit.clear(); dnaPacked.clear(); dnaBit.clear(); dnaInfo.clear(); nodes.clear(); topNodes.clear(); emRecs.clear(); trees.clear(); bins.clear(); }
};

}  // namespace fs
