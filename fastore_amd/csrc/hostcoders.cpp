// Host build of the entropy-coder cores (one "lane"), used ONLY for the merged small-bins / N
// block ("block 0"): a single serial PPMd stream of up to hundreds of MB per archive
// (RawCompressorSE/PE, /root/reference/fastore/fastore_pack/FastqCompressor.cpp:3407-3890) that
// one 2.4 GHz wavefront would take minutes to walk.  Standard bins never come through here.
#include "hostcoders.h"
#include <stdlib.h>
#include <stdexcept>
#include "ppmd_core.h"
#include "rc_core.h"
#include "qvz_core.h"

namespace fshost {

void ppmdEncode(const uint8_t* in, size_t n, std::vector<uint8_t>& out)
{
    out.clear();
    if (n == 0) return;
    if (n > 0xFFFFFF00ull) throw std::runtime_error("block-0 stream exceeds 4 GiB");
    out.resize(n + n / 8 + 1024);
    uint8_t* arena = (uint8_t*)aligned_alloc(64, (fsppmd::ARENA_BYTES + 63) & ~63ull);
    fsppmd::Shared* sh = new fsppmd::Shared;
    const uint32_t sz = fsppmd::encode_member(arena, sh, in, (uint32_t)n, out.data(), (uint32_t)out.size(), nullptr);
    delete sh; free(arena);
    if (sz >= out.size()) throw std::runtime_error("block-0 PPMd output overflow");
    out.resize(sz);
}

void rcEncode(uint32_t model, const uint8_t* pairs, size_t nPairs, std::vector<uint8_t>& out)
{
    if (nPairs > 0x7FFFFFF0ull) throw std::runtime_error("block-0 range-coded stream too long");
    const uint64_t tb = fsrc::model_table_bytes(model);
    uint8_t* table = (uint8_t*)aligned_alloc(64, (tb + 63) & ~63ull);
    out.resize(2 * nPairs + 64);
    const uint32_t sz = fsrc::encode_model(model, table, pairs, (uint32_t)nPairs, out.data(), (uint32_t)out.size());
    free(table);
    out.resize(sz);
}

void qvzEncode(const uint8_t* modelBlob, const uint8_t* symbols, size_t nSymbols, std::vector<uint8_t>& out)
{
    if (nSymbols > 0x3FFFFFF0ull) throw std::runtime_error("block-0 QVZ stream too long");
    fsqvz::ModelHeader h; memcpy(&h, modelBlob, sizeof h);
    uint8_t* arena = (uint8_t*)aligned_alloc(64, (4ull * h.image_words + 127) & ~63ull);
    out.resize(3 * nSymbols + 64);
    const uint32_t sz = fsqvz::encode_stream(arena, modelBlob, symbols, (uint32_t)nSymbols, out.data(), (uint32_t)out.size());
    free(arena);
    if (sz == 0xFFFFFFFFu) throw std::runtime_error("block-0 QVZ stream: malformed symbol or output overflow");
    out.resize(sz);
}

}  // namespace fshost
