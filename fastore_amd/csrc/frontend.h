// Per-bin read-cluster modelling ("front end" of the hot path, SURVEY §8 a3-a6, a8, a10, a11,
// a14): sorts the bin, builds the LZ match tree, the contigs, walks the trees in the
// reference's order and emits the work streams that the device entropy-codes.
//
// Round-1 placement: this stage runs on host cores (one bin per task, independent bins) and feeds
// the HIP kernels; DESIGN.md lists it as the next stage to move on-device.  Every function cites
// the reference code whose decisions it reproduces bit-for-bit.
#pragma once
#include <stdint.h>
#include <deque>
#include <vector>
#include <functional>
#include <memory>
#include "format.h"
#include "qvz.h"
#include "device_types.h"

namespace fs {

struct PackParams {                       // fastore_pack CLI knobs (fastore_pack/Params.h:18-147)
    uint32_t minBinSize = 256;            // -f
    int32_t encodeThreshold = 0;          // -e (0 = auto: seqLen / 2)
    int32_t pairEncodeThreshold = 0;      // -E
    int32_t shiftCost = 1;                // -s
    int32_t mismatchCost = 2;             // -m
    uint32_t maxLzWindowSize = 255;       // -w  (MAX_LZ_SE)
    uint32_t maxPairLzWindowSize = 255;   // -W  (MAX_LZ_PE)
    bool extraReduceHardReads = false;    // -r
    bool extraReduceExpensiveLzMatches = false;   // -l
    uint32_t beginCut = 2, endCut = 2;
    uint32_t maxNewVariantsPerRead = 1;   // -n
    uint32_t maxRecordShiftDifference = 0;   // -q
    uint32_t maxHammingDistance = 8;      // -d
    uint32_t minConsensusSize = 10;       // -c
    uint32_t maxMismatchesLowCost = 4;
};

// stream indices: fastore_pack/CompressedBlockData.h:97-154
enum Stream {
    S_Flag = 0, S_LettersX, S_Rev, S_HardReads, S_LzId, S_Shift, S_Match, S_MatchBinary, S_TreeShift, S_CMatch, S_CShift, S_CLetters,
    S_Quality, S_IdToken, S_IdValue, S_SE_COUNT,
    S_FlagPE = S_SE_COUNT, S_LettersXPE, S_SwapPE, S_HardPE, S_LzIdPE, S_ShiftPE, S_MatchRlePE, S_MatchBinaryPE, S_PE_COUNT
};

// one quality string of the device-side quality path, in emission order: where its scores start in the bin's packed
// .bqua bytes, how many there are, and whether the string is emitted back to front (IQualityStoreBase::CompressReadQuality,
// FastqCompressor.cpp:236: a reverse-complemented record's scores are coded in the read's original orientation)
// (8-bin / binary archives: positions under an 'N' base are not coded, FastqCompressor.cpp:277, 307 -- nCount of them,
// as stored indices, from quaN[nFirst] on)
struct QuaRef { uint32_t bit; uint16_t len; uint8_t reverse, nCount; uint32_t nFirst; };

// one read id of the device-side tokeniser, in emission order: where its stored characters start in the bin's packed
// .bhead bytes, and its length (the leading '@' is implied)
struct IdRef { uint32_t bit; uint32_t len; };

struct BinStreams {
    std::vector<uint8_t> s[S_PE_COUNT];   // P streams: raw bytes; R streams: (symbol, ctx0) byte pairs
    uint64_t rawIdSize = 0;
    uint32_t nStreams = S_SE_COUNT;
    // device-side quality path: s[S_Quality] stays empty, the stream is these strings one after the other
    std::vector<QuaRef> quaRefs; uint64_t quaSymbols = 0; std::vector<uint8_t> quaN;
    const uint8_t* quaPacked = nullptr; uint64_t quaPackedBytes = 0;      // the bin's .bqua bytes (owned by the batch)
    // device-side read-id tokeniser: s[S_IdToken] / s[S_IdValue] stay empty, the device writes them from these
    std::vector<IdRef> idRefs; const uint8_t* headPacked = nullptr; uint64_t headPackedBytes = 0;
    // device-side emission (fsdev::EmitOp, emit_core.h): the streams that hold bases -- HardReads, LettersX, Match, MatchBinary, CMatch,
    // CLetters, their paired-end counterparts -- and LzId stay empty; the device writes them from the ops the walk left, the bin's
    // bases [emitSeqLo, emitSeqHi) of the batch's base array, its contigs' bytes and its LZ ids
    bool deviceEmit = false;
    bool pairsPending = false;             // the bin's pairs went to the device's batched mate search: its mate streams are still to come
    std::vector<fsdev::EmitOp> emitOps; std::vector<uint32_t> lzIds; std::vector<uint8_t> contigBytes;
    uint64_t emitSeqLo = 0, emitSeqHi = 0; const uint8_t* emitSeq = nullptr;      // (emitSeq: the first of those bases; owned by the batch)
    uint64_t emitBound[fsdev::ECH_COUNT + 1] = {0};          // per channel: an upper bound of its units (bytes, pairs, bits); [ECH_COUNT]: LzId bytes
    void reset(uint32_t n) { nStreams = n; rawIdSize = 0; for (auto& v : s) v.clear(); quaRefs.clear(); quaSymbols = 0; quaN.clear(); quaPacked = nullptr; quaPackedBytes = 0; idRefs.clear(); headPacked = nullptr; headPackedBytes = 0;
                             deviceEmit = false; pairsPending = false; emitOps.clear(); lzIds.clear(); contigBytes.clear(); emitSeqLo = emitSeqHi = 0; emitSeq = nullptr; for (auto& b : emitBound) b = 0; }
};

// which streams are range-coded in place (true) vs PPMd-compressed (false), and with which model
// (ILzCompressorBase::SetupBufferMask, FastqCompressor.h:557-581; PE: FastqCompressor.cpp:4309-4319)
bool streamIsRangeCoded(uint32_t stream, uint32_t qualityMethod);
uint32_t streamModel(uint32_t stream, uint32_t qualityMethod);

// archive-level parameters of one library (they travel inside .bmeta)
struct ArchiveParams { BinModuleConfigRaw cfg{}; HeaderStats head; QvzModel qvz; };

// The device-side window search (matcher.hip) as the front end sees it: the bin's bases, the table of its match-tree
// constructions, one row of answers per read.  Returns false when the search could not be run (the host scan is used).
// packed != nullptr: the same bases as the bin file stores them (fsdev::PackedDna): the device unpacks them itself and `seq` does not travel.
typedef std::function<bool(const uint8_t* seq, size_t seqBytes, const fsdev::PackedDna* packed, const fsdev::MatchRead* reads, size_t nReads, const fsdev::MatchCall* calls,
                           size_t nCalls, const uint32_t* warm, size_t nWarm, const fsdev::MatchParams& par, fsdev::MatchRow* rows)> MatchFn;

// The device-side mate search of a paired-end bin (matcher.hip: fs_match_mates): the bin's bases, its pairs in the order the
// tree walk emits them, the archive's valid signatures (a bit each); one row per pair.  false: could not be run (host search).
typedef std::function<bool(const uint8_t* seq, size_t seqBytes, const fsdev::MatePair* pairs, size_t nPairs, const uint32_t* validBits, size_t validWords,
                           const fsdev::MateParams& par, fsdev::MateRow* rows)> MateFn;

// A paired-end bin whose mate searches are the device's AND are not waited for: the walk is over, the host thread goes on to its
// next bin, and what is left of the bin -- every pair's search (fs_match_mates, many bins per launch) and, from the rows, the mate's
// streams -- happens when the batch the search rides in comes back (emitPairsFromRows).  Everything that step needs stands here: the
// front end's per-thread state is long busy with another bin.  (Only with device-side emission: the mate's streams are then ops.)
struct PendingPairs {
    BinStreams* out = nullptr;
    std::vector<fsdev::MatePair> pairs;            // in the walk's order; mate_off counts from `seq`
    std::vector<uint8_t> seType;                   // the single-end match type of each pair's first read (the PE flag's context)
    std::vector<fsdev::MateRow> rows;              // the searches' answers (filled by the device)
    const uint8_t* seq = nullptr; size_t seqBytes = 0;     // the mates' bases: one stretch of the batch's base array ...
    uint64_t seqLo = 0;                            // ... beginning at this offset of it
    fsdev::MateParams mp{};
    std::vector<uint32_t> validBits;               // the valid-signature table of the archive's minimizer parameters
    int32_t shiftCost = 1, mismatchCost = 2; uint32_t maxMismatchesLowCost = 0;
    std::function<void(const char* error)> done;   // the bin is complete (or failed)
};
typedef std::function<bool(std::unique_ptr<PendingPairs>)> AsyncMateFn;     // true: taken (done() will be called, possibly before this returns)
// the mate streams of a bin from its searches' rows (LzCompressorPE::CompressPair's coding half, FastqCompressor.cpp:4740-4900), as ops
void emitPairsFromRows(PendingPairs& pp);

class BinEncoder {
public:
    // paired-end bins with device-side emission: hand the pairs over instead of searching (empty: search now)
    void setAsyncMates(AsyncMateFn fn);
    explicit BinEncoder(const PackParams& par);
    // mate searches of the following encodeLz calls (paired-end bins) go through `fn` (empty: the host search)
    void setMateMatcher(MateFn fn);
    // parity check of the device mate search: the bin through the host search and through `fn`; pairs, and rows that differ
    void checkMateMatcher(const Batch& data, const Batch& graph, const BinIn& bin, const ArchiveParams& arch, const MateFn& fn, uint64_t& pairs, uint64_t& differing);
    // window searches of the following encodeLz calls go through `fn` (empty: the host scan)
    void setMatcher(MatchFn fn);
    // asked once per bin before anything is made for the device search: false = this bin's searches are the host's scan (the packer's
    // answer for the bins behind the heaviest ones: "only while fewer searches are waiting for the device than threads may wait for free")
    void setMatcherGate(std::function<bool()> fn);
    // the streams that hold bases are written by the device from ops (fsdev::EmitOp) instead of by the walk itself
    void setDeviceEmit(bool on);
    // parity check of the device matcher: runs the bin through the host scan and through `fn`, returns the number of reads
    // and of rows that differ (match, cost, shift, mismatch-free flag, duplicate flag)
    void checkMatcher(const Batch& data, const Batch& graph, const BinIn& bin, const ArchiveParams& arch, const MatchFn& fn, uint64_t& reads, uint64_t& differing);
    // standard bin: LzCompressorSE/PE::Compress up to (not including) CompressBuffers
    void encodeLz(const Batch& batch, const BinIn& bin, const ArchiveParams& arch, BinStreams& out);
    // the same with the bin's stored graph in a batch of its own (node indices local to `graph`, record indices into `data`)
    void encodeLz(const Batch& data, const Batch& graph, const BinIn& bin, const ArchiveParams& arch, BinStreams& out);

private:
    struct Impl;
    Impl* impl_;
public:
    ~BinEncoder();
    BinEncoder(const BinEncoder&) = delete;
    BinEncoder& operator=(const BinEncoder&) = delete;
};

// read-id tokeniser shared with the raw (block 0) coder: IHeaderStoreBase::CompressReadId
// (fastore_pack/FastqCompressor.cpp:504-583).  Appends (symbol, ctx) pairs.
struct FieldSpec { uint8_t method; };     // 0 const, 1 token, 2 raw numeric
void compressReadId(const HeaderStats& head, const uint8_t* h, uint32_t headLen, std::vector<uint8_t>& tokenPairs,
                    std::vector<uint8_t>& valuePairs);
uint32_t intLog(uint64_t x, uint64_t base);       // int_log of fastore_bin/Utils.h as CompressReadId uses it (bytes of a numeric field)
// IQualityStoreBase::CompressReadQuality (FastqCompressor.cpp:221-364); MET_QVZ needs the library's model and the
// block's running WELL generator (reset at every block start, FastqCompressor.cpp:906-915)
void compressReadQuality(const BinModuleConfigRaw& cfg, const uint8_t* seq, const uint8_t* qua, uint32_t len, bool reverse,
                         std::vector<uint8_t>& out, const QvzModel* qvz = nullptr, WellRng* rng = nullptr);

}  // namespace fs
