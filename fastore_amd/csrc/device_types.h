// Plain descriptors shared by the host engine and the HIP kernels.
#pragma once
#include <stdint.h>

namespace fsdev {

enum : uint32_t { KIND_PPMD = 0, KIND_RC_BASE = 1 /* + fsrc::Model */, KIND_QVZ = 64 /* fsqvz arithmetic coder */ };

// one entropy-coded stream of one bin
struct StreamItem {
    uint64_t in_off;      // byte offset into the batch input buffer (even for RC pair streams)
    uint64_t out_off;     // byte offset into the batch scratch-output buffer
    uint32_t in_len;      // bytes (PPMd), (symbol, ctx) pairs (RC) or u32 symbols (QVZ)
    uint32_t out_cap;     // bytes available at out_off
    uint32_t kind;        // KIND_PPMD, KIND_RC_BASE + model, or KIND_QVZ
    uint32_t bin;         // bin index inside the batch
    uint64_t aux_off;     // KIND_QVZ: byte offset of the library's model blob in the batch input buffer (16-byte aligned)
};

enum : uint32_t { MAX_STREAMS = 23 };

// per-bin block assembly plan (reference block layout: SURVEY §8 a13,
// /root/reference/fastore/fastore_pack/FastqCompressor.cpp:56-69, 684-699, 1055-1126, 1199-1210)
struct BlockPlan {
    uint64_t block_off;               // offset of the block in the compact output buffer
    uint64_t records;                 // header fields
    uint64_t raw_dna_size;
    uint64_t raw_id_size;
    uint32_t signature;
    uint32_t n_streams;               // 15 (SE) or 23 (PE)
    uint32_t first_item;              // items [first_item, first_item + n_streams) in STREAM order
    uint8_t min_len, max_len, has_headers, pad;
    uint32_t copy_order[MAX_STREAMS]; // stream indices in the order their bytes are laid out
    uint64_t work_size[MAX_STREAMS];  // pre-entropy ("work") sizes for the header
};

}  // namespace fsdev
