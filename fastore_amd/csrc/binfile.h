// Reader for the bin/rebin -> pack hand-off files and the record/graph unpacker (drop-in boundary,
// SURVEY §8b "Input").  Restates, with flat SoA output:
//   BinFileReader::StartDecompress/ReadFileFooter/ReadBlock   fastore_bin/BinFile.cpp:470-813
//   BinFileExtractor (std / small / N split)                   fastore_pack/BinFileExtractor.cpp:21-103
//   IFastqNodesPacker::UnpackFromBin / ReadNextNode            fastore_rebin/NodesPacker.cpp:416-679
//   FastqNodesPackerSE/PE::ReadRecordData / ReadExactMatch     fastore_rebin/NodesPacker.cpp:705-979
//   IFastqPacker::ReadNextRecord / ReadDna / ReadQuality / ReadHeader   fastore_bin/FastqPacker.cpp:62-411
#pragma once
#include <map>
#include <string>
#include <vector>
#include "format.h"
#include "qvz.h"

namespace fs {

class BinFile {
public:
    ~BinFile();
    void open(const std::string& prefix, uint32_t minBinSize);
    void close();

    const BinModuleConfigRaw& config() const { return cfg_; }
    const HeaderStats& headData() const { return head_; }
    const QvzModel& qvz() const { return qvz_; }
    const std::map<uint32_t, BinInfo>& bins() const { return bins_; }
    const std::vector<uint32_t>& stdSignatures() const { return std_; }
    const std::vector<uint32_t>& smallSignatures() const { return small_; }
    bool hasNBin() const { return bins_.count(nSignature()) != 0; }
    uint32_t nSignature() const { return 1u << (2 * cfg_.minimizer.signatureLen); }
    bool usesHeaders() const { return usesHeaderStream_; }

    // Unpack one signature into `batch`.  asNewBin: start a new BinIn; otherwise append the
    // records/nodes to the last bin of the batch (block-0 merge of small bins and the N bin).
    // Thread-safe: only reads the mapped files and the footer tables.
    // keepPackedDna (new bins only): the bin's .bdna bytes and every record's place in them are kept too (Batch::dnaPacked / dnaBit / dnaInfo)
    void unpack(uint32_t signature, Batch& batch, bool asNewBin, bool keepPackedDna = false) const;
    // Placed form for parallel batch assembly: bases/quals/headers/records go to pre-sized arrays of `data` at the given
    // offsets (their sizes are known from the footer); the graph tables and the BinIn go to `graph` (node indices local to it).
    // quaBase >= 0: the qualities are not unpacked; the bin's .bqua bytes go to data.quaPacked at quaBase and every record's
    // bit offset into them to data.quaBit (lossless archives only; see Batch)
    // headPackedBase >= 0: the same for the read ids (data.headPacked / data.headBit)
    void unpackPlaced(uint32_t signature, Batch& data, uint64_t seqBase, uint64_t headBase, uint32_t recBase, Batch& graph, int64_t quaBase = -1, int64_t headPackedBase = -1, int64_t dnaPackedBase = -1) const;

private:
    void readFooter(const std::vector<uint8_t>& buf);
    void unpackImpl(uint32_t signature, Batch& data, Batch& graph, bool asNewBin, bool placed, uint64_t seqBase, uint64_t headBase, uint32_t recBase, int64_t quaBase, int64_t headPackedBase, int64_t dnaPackedBase) const;
    struct Map { const uint8_t* p = nullptr; uint64_t size = 0; };    // read-only mmap of one stream file
    static Map mapFile(const std::string& name);
    static void unmap(Map& m);
    static void copyAt(const Map& m, uint64_t off, void* dst, uint64_t n, const char* what);

    Map meta_, dna_, qua_, headf_;
    BinModuleConfigRaw cfg_{};
    bool usesHeaderStream_ = false;
    std::map<uint32_t, BinInfo> bins_;
    HeaderStats head_;
    QvzModel qvz_;
    std::vector<uint32_t> std_, small_;
};

}  // namespace fs
