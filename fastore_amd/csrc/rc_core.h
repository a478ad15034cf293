// Order-k adaptive range coders, one stream per wavefront  (SURVEY §8 a7).
//
// Bit-exact restatement of the reference coder stack for the streams fastore_pack codes "in
// place" (Rev, MatchBinary, LettersX, CLetters, read-id tokens/values, binary / 8-bin quality,
// PE flag):
//   RangeEncoder          /root/reference/fastore/rc/RangeCoder.h:40-84
//   TSymbolCoderRC        /root/reference/fastore/rc/SymbolCoderRC.h:19-93
//   TSimple/TAdvancedContextCoder   /root/reference/fastore/rc/ContextEncoder.h:84-206
//   TEncoder::Start/End   /root/reference/fastore/rc/ContextEncoder.h:208-250
//
// MI355X mapping: the uint16 frequency table of a stream (up to 32 MiB for the <256,1> and <8,6>
// models) sits in the wave's private HBM arena and is initialised by the wave itself with
// 16-byte stores; for the 256-symbol alphabet every lane owns four symbols, so the
// O(alphabet) accumulate / cumulative-frequency steps of the reference become one 8-byte load
// per lane plus a wave prefix scan.  Small alphabets (2, 8) are one 16-byte uniform load.
#pragma once
#include "wave.h"

namespace fsrc {

struct Enc {
    uint64_t low; uint32_t range;
    fs_gptr out; uint32_t cap, pos;
};

FS_DEV void put(Enc& e, uint32_t b) { if (e.pos < e.cap) fs_st8(e.out + e.pos, b); e.pos++; }

FS_DEV void encode_freq(Enc& e, uint32_t symFreq, uint32_t cumFreq, uint32_t total)
{
    e.range /= total;
    e.low += (uint32_t)(e.range * cumFreq);
    e.range *= symFreq;
    while (e.range <= 0x00ffffffu) {
        if ((e.low ^ (e.low + e.range)) & 0xff00000000000000ULL) {
            const uint32_t x = (uint32_t)e.low;
            e.range = (x | 0x00ffffffu) - x;
        }
        put(e, (uint32_t)(e.low >> 56));
        e.low <<= 8; e.range <<= 8;
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
// inclusive wave scan of one value per lane; returns exclusive prefix, *total = wave sum
FS_DEV uint32_t wave_excl_scan(uint32_t v, uint32_t* total)
{
    uint32_t x = v;
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d, 64); if (FS_LANE() >= d) x += y; }
    *total = __shfl(x, 63, 64);
    return x - v;
}
FS_DEV uint32_t wave_bcast(uint32_t v, int lane) { return __shfl(v, lane, 64); }
#else
FS_DEV uint32_t wave_excl_scan(uint32_t v, uint32_t* total) { *total = v; return 0; }
FS_DEV uint32_t wave_bcast(uint32_t v, int) { return v; }
#endif

// table bytes needed for a model
FS_DEV uint64_t table_bytes(int bits, int order, int adv) { return (1ULL << (bits * (order + (adv ? 1 : 0)))) * (1ULL << bits) * 2ULL; }

// Encode n (symbol, ctx0) byte pairs; returns the stream size including the 8 flush bytes
// (counted past `cap` like the reference's growing writer would; bytes beyond cap are dropped).
// CTXBITS: width of the ctx0 slot in the model index.  The reference uses BITS (ContextEncoder.h:184); a narrower slot
// addresses the same models in a denser table when every ctx0 of the stream is known to fit (read-id streams: ctx0 =
// fieldId*4+k < 64), which is only a change of memory layout.
template <int BITS, int ORDER, bool ADV, int CTXBITS = BITS>
FS_DEV uint32_t encode_stream(fs_gptr table /*16-byte aligned, table_bytes()*/, fs_cgptr pairs, uint32_t n,
                              fs_gptr out, uint32_t cap)
{
    constexpr uint32_t A = 1u << BITS;
    constexpr uint64_t symMask = (1ULL << (ORDER * BITS)) - 1;
    constexpr uint64_t nModels = 1ULL << (BITS * ORDER + (ADV ? CTXBITS : 0));
    constexpr uint32_t limit = (1u << 16) - A * 8u;
    // Clear(): every statistic = 1
    {
        const uint64_t words = nModels * A / 2;                     // u32 words of two u16 ones
        fs_gptr32 t32 = (fs_gptr32)table;
        if (words >= 4u * FS_WAVE) {
            struct alignas(16) V4 { uint32_t a, b, c, d; };
            FS_GLOBAL V4* t128 = (FS_GLOBAL V4*)table; const V4 ones = {0x00010001u, 0x00010001u, 0x00010001u, 0x00010001u};
            for (uint64_t i = (uint64_t)FS_LANE(); i < words / 4; i += FS_WAVE) t128[i] = ones;
        } else {
            for (uint64_t i = (uint64_t)FS_LANE(); i < words; i += FS_WAVE) t32[i] = 0x00010001u;
        }
        FS_WAVE_SYNC();
    }
    Enc e; e.low = 0; e.range = 0xffffffffu; e.out = out; e.cap = cap; e.pos = 0;
    uint64_t hash = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t pr = fs_ld16(pairs + 2u * k);
        const uint32_t sym = pr & 0xFFu, ctx = pr >> 8;
        // a symbol outside the alphabet has no statistic (its frequency would read as 0 and the normalisation loop below
        // would never end); a context outside its field would index past the table.  Both only come from corrupted input
        // or a wrong caller: give the stream up (reported as 0xFFFFFFFF, like the QVZ coder does).
        if ((BITS < 8 && sym >= A) || (ADV && CTXBITS < 8 && ctx >= (1u << CTXBITS))) return 0xFFFFFFFFu;
        const uint32_t h = ADV ? (uint32_t)(((hash & symMask) << CTXBITS) | ctx) : (uint32_t)(hash & symMask);
        fs_gptr16 st = (fs_gptr16)table + (uint64_t)h * A;
        uint32_t acc, lo, f;
        if constexpr (BITS == 8) {
            constexpr int PER = 256 / FS_WAVE;                      // symbols owned by one lane
            const int lane = FS_LANE();
            uint32_t v[PER]; uint32_t s = 0;
            for (int j = 0; j < PER; ++j) { v[j] = st[lane * PER + j]; s += v[j]; }
            uint32_t ex = wave_excl_scan(s, &acc);
            if (acc >= limit) {                                      // Rescale(): stats -= stats >> 1
                s = 0;
                for (int j = 0; j < PER; ++j) { v[j] -= v[j] >> 1; st[lane * PER + j] = (uint16_t)v[j]; s += v[j]; }
                ex = wave_excl_scan(s, &acc);
            }
            const int owner = (int)(sym / PER), within = (int)(sym % PER);
            uint32_t myLo = ex, myF = 0;
            for (int j = 0; j < PER; ++j) { if (j < within) myLo += v[j]; if (j == within) myF = v[j]; }
            lo = FS_UNI(wave_bcast(myLo, owner)); f = FS_UNI(wave_bcast(myF, owner)); acc = FS_UNI(acc);
            if (lane == owner) st[sym] = (uint16_t)(f + 8);      // lanes only ever touch their own 4 symbols: no sync
        } else {
            // the model's A statistics: lane j fetches statistic j (one instruction), broadcast to scalars
            uint32_t v[A]; acc = 0;
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t mine = (uint32_t)FS_LANE() < A ? (uint32_t)st[FS_LANE()] : 0u;
            for (uint32_t j = 0; j < A; ++j) { v[j] = fs_readlane(mine, j); acc += v[j]; }
            if (acc >= limit) {
                acc = 0;
                for (uint32_t j = 0; j < A; ++j) { v[j] -= v[j] >> 1; acc += v[j]; }
                if ((uint32_t)FS_LANE() < A) st[FS_LANE()] = (uint16_t)(mine - (mine >> 1));
            }
#else
            for (uint32_t j = 0; j < A; ++j) { v[j] = st[j]; acc += v[j]; }
            if (acc >= limit) {
                acc = 0;
                for (uint32_t j = 0; j < A; ++j) { v[j] -= v[j] >> 1; st[j] = (uint16_t)v[j]; acc += v[j]; }
            }
#endif
            lo = 0; f = 0;
            for (uint32_t j = 0; j < A; ++j) { if (j < sym) lo += v[j]; if (j == sym) f = v[j]; }
            fs_st16((fs_gptr)(st + sym), f + 8);
        }
        encode_freq(e, f, lo, acc);
        hash = (hash << BITS) | sym;
    }
    for (int i = 0; i < 8; ++i) { put(e, (uint32_t)(e.low >> 56)); e.low <<= 8; }
    return e.pos;
}

// model ids shared by host and device
enum Model : uint32_t { M_S2O4 = 0, M_S8O4 = 1, M_A8O4 = 2, M_A2O10 = 3, M_A8O6 = 4, M_A256O1 = 5, M_A256O1_C6 = 6 /* ctx0 < 64: 8 MiB table */, M_COUNT = 7 };

FS_DEV uint32_t encode_model(uint32_t model, fs_gptr table, fs_cgptr pairs, uint32_t n, fs_gptr out, uint32_t cap)
{
    switch (model) {
    case M_S2O4: return encode_stream<1, 4, false>(table, pairs, n, out, cap);
    case M_S8O4: return encode_stream<3, 4, false>(table, pairs, n, out, cap);
    case M_A8O4: return encode_stream<3, 4, true>(table, pairs, n, out, cap);
    case M_A2O10: return encode_stream<1, 10, true>(table, pairs, n, out, cap);
    case M_A8O6: return encode_stream<3, 6, true>(table, pairs, n, out, cap);
    case M_A256O1_C6: return encode_stream<8, 1, true, 6>(table, pairs, n, out, cap);
    default: return encode_stream<8, 1, true>(table, pairs, n, out, cap);
    }
}
FS_DEV uint64_t model_table_bytes(uint32_t model)
{
    switch (model) {
    case M_S2O4: return table_bytes(1, 4, 0);
    case M_S8O4: return table_bytes(3, 4, 0);
    case M_A8O4: return table_bytes(3, 4, 1);
    case M_A2O10: return table_bytes(1, 10, 1);
    case M_A8O6: return table_bytes(3, 6, 1);
    case M_A256O1_C6: return (1ULL << 14) * 512ULL;
    default: return table_bytes(8, 1, 1);
    }
}
}  // namespace fsrc
