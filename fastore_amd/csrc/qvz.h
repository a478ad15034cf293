// QVZ (--lossy) quality model of one library: the conditional-quantizer codebook and the WELL-1024a seed
// that travel in the .bmeta footer, flattened for the front end (symbolisation) and for the device coder.
//
// Reference: codebook serialisation fastore_bin/QVZ.cpp:165-302 (WriteCodebook / ReadCodebook), footer
// placement fastore_bin/BinFile.cpp:386-394, 740-755 and fastore_pack/ArchiveFile.cpp:134-146, quantizer
// choice fastore_pack/quantizer.cpp:522-531, alphabets fastore_pack/pmf.cpp:20-46, 314-396 and
// quantizer.cpp:449-480, PRNG fastore_pack/well.cpp:16-57, per-read loop FastqCompressor.cpp:318-364.
#pragma once
#include <stdint.h>
#include <memory>
#include <mutex>
#include <vector>
#include "bitio.h"

namespace fs {

enum : uint32_t { QVZ_ALPHABET = 72, QVZ_NOT_SYMBOL = 127, QVZ_INDEX_SLOTS = 82 /* ALPHABET_SIZE + 10 */ };

// WELL-1024a with the reference's bit-amortising front (well.cpp:16-57)
struct WellRng {
    uint32_t state[32]; uint32_t n = 0, bitOutput = 0, bitsLeft = 0;
    void reset(const uint32_t* seed) { for (int i = 0; i < 32; ++i) state[i] = seed[i]; n = 0; bitOutput = 0; bitsLeft = 0; }
    uint32_t next()
    {
        uint32_t* s = state;
        const uint32_t z0 = s[(n + 31) & 31], vm1 = s[(n + 3) & 31], vm2 = s[(n + 24) & 31], vm3 = s[(n + 10) & 31];
        const uint32_t z1 = s[n] ^ (vm1 ^ (vm1 >> 8));
        const uint32_t z2 = (vm2 ^ (vm2 << 19)) ^ (vm3 ^ (vm3 << 14));
        s[n] = z1 ^ z2;
        n = (n + 31) & 31;
        s[n] = (z0 ^ (z0 << 11)) ^ (z1 ^ (z1 << 7)) ^ (z2 ^ (z2 << 13));
        return s[n];
    }
    uint32_t bits(uint32_t k)
    {
        if (bitsLeft < k) { bitOutput = next(); bitsLeft = 32; }
        const uint32_t r = bitOutput & ((1u << k) - 1);
        bitOutput >>= k; bitsLeft -= k;
        return r;
    }
};

struct QvzModel {
    bool present = false;
    uint32_t wellSeed[32] = {};
    uint32_t maxReadLength = 0;                 // = number of codebook columns
    std::vector<uint8_t> footerBytes;           // WELL state + max_read_length + codebook, verbatim (re-emitted in .cmeta)

    // one "context" = one quantizer = (column, previous-value index, low/high)
    uint32_t nCtx = 0;
    std::vector<uint32_t> colCtxBase;           // [columns] first context of the column
    std::vector<uint16_t> colIndex;             // [columns][QVZ_INDEX_SLOTS] previous quantized value -> index in the column's input alphabet, 0xFFFF absent
    std::vector<uint8_t> qratio;                // [nCtx / 2] threshold of the low/high draw
    std::vector<uint8_t> quant;                 // [nCtx][QVZ_ALPHABET] quality value -> quantized value
    std::vector<uint8_t> stateOf;               // [nCtx][QVZ_ALPHABET] quality value -> index of the quantized value in the output alphabet
    std::vector<uint8_t> card;                  // [nCtx] output alphabet size

    // device blob (fsqvz::ModelHeader, descriptors, initial statistics image): see qvz_core.h
    std::vector<uint8_t> blob;
    // the quantizer tables as fs_gather_quality_qvz reads them (fsdev::QvzSymHeader + tables, WITHOUT the generator's words)
    std::vector<uint8_t> symBlob;
    // fsdev::QvzSymHeader | tables | the generator's first `wellWords` outputs, into dst (symBlobBytes(wellWords) bytes).  The
    // outputs are the same for every bin of the archive (the generator is re-seeded per bin): made once, kept, extended on demand
    size_t symBlobBytes(uint32_t wellWords) const { return ((symBlob.size() + 15u) & ~(size_t)15u) + 4ull * wellWords; }
    void writeSymBlob(uint8_t* dst, uint32_t wellWords) const;

    // parses the footer section at the reader's position (byte aligned) and builds everything above
    void parse(BitReader& r);

private:
    struct WellCache { std::mutex mx; WellRng rng; std::vector<uint32_t> words; };
    std::shared_ptr<WellCache> well_;           // shared by the copies of one archive's model
};

// FastqCompressor.cpp:318-364: one read's qualities -> one u32 per position (context | state << 24)
void qvzSymbolise(const QvzModel& m, WellRng& rng, const uint8_t* qua, uint32_t len, uint32_t qualityOffset, bool reverse, std::vector<uint8_t>& out);

}  // namespace fs
