/* TEST INFRASTRUCTURE (oracle) -- CPU restatement, never linked into the product.
 *
 * Order-k adaptive range coders of fastore_pack and the two RLE byte emitters:
 *   RangeEncoder            fastore/rc/RangeCoder.h:40-84   (64-bit low, 32-bit range, carry-less)
 *   TSymbolCoderRC          fastore/rc/SymbolCoderRC.h:19-93 (uint16 stats, init 1, +8, halve at limit)
 *   TSimpleContextCoder     fastore/rc/ContextEncoder.h:155-175  model = hash & mask
 *   TAdvancedContextCoder   fastore/rc/ContextEncoder.h:176-206  model = ((hash & mask) << bits) | ctx0
 *   BinaryRleEncoder        fastore/rle/RleEncoder.h:21-79
 *   Rle0Encoder             fastore/rle/RleEncoder.h:140-212
 * Parity pin: tests/test_oracle_rc.py against oracle/_ref/ref_driver rc and committed vectors.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t low; uint32_t range; uint8_t* out; size_t cap, pos; } rc_t;

static void rc_put(rc_t* r, uint32_t b) { if (r->pos < r->cap) r->out[r->pos] = (uint8_t)b; r->pos++; }

static void rc_encode_freq(rc_t* r, uint32_t symFreq, uint32_t cumFreq, uint32_t total)   /* RangeCoder.h:53-71 */
{
    r->range /= total;
    r->low += (uint32_t)(r->range * cumFreq);
    r->range *= symFreq;
    while (r->range <= 0x00ffffffu) {
        if ((r->low ^ (r->low + r->range)) & 0xff00000000000000ULL) {
            uint32_t x = (uint32_t)r->low;
            r->range = (x | 0x00ffffffu) - x;
        }
        rc_put(r, (uint32_t)(r->low >> 56));
        r->low <<= 8; r->range <<= 8;
    }
}

/* model: alphabet = 1<<bits symbols, symbol order `order`, advanced = ctx0 appended */
size_t fso_rc_encode(int bits, int order, int advanced, const uint8_t* sym, const uint8_t* ctx, size_t n,
                     uint8_t* out, size_t cap)
{
    const uint32_t A = 1u << bits;
    const uint64_t symMask = (1ULL << (order * bits)) - 1;
    const uint64_t nModels = 1ULL << (bits * (order + (advanced ? 1 : 0)));
    const uint32_t limit = (1u << 16) - A * 8;                        /* SymbolCoderRC.h:64 */
    uint16_t* stats = (uint16_t*)malloc(nModels * A * sizeof(uint16_t));
    for (uint64_t i = 0; i < nModels * A; ++i) stats[i] = 1;          /* Clear(): ContextEncoder.h:112-120 */
    rc_t r = {0, 0xffffffffu, out, cap, 0};                           /* Start(): RangeCoder.h:47-51 */
    uint64_t hash = 0;
    for (size_t k = 0; k < n; ++k) {
        uint32_t h = advanced ? (uint32_t)(((hash & symMask) << bits) | ctx[k]) : (uint32_t)(hash & symMask);
        uint16_t* st = stats + (uint64_t)h * A;
        uint32_t acc = 0;                                             /* Accumulate(): SymbolCoderRC.h:74-90 */
        for (uint32_t i = 0; i < A; ++i) acc += st[i];
        if (acc >= limit) {
            acc = 0;
            for (uint32_t i = 0; i < A; ++i) st[i] = (uint16_t)(st[i] - (st[i] >> 1));
            for (uint32_t i = 0; i < A; ++i) acc += st[i];
        }
        uint32_t lo = 0;
        for (uint32_t i = 0; i < sym[k]; ++i) lo += st[i];
        rc_encode_freq(&r, st[sym[k]], lo, acc);                      /* SymbolCoderRC.h:30-43 */
        st[sym[k]] = (uint16_t)(st[sym[k]] + 8);
        hash = (hash << bits) | sym[k];                               /* UpdateHash: ContextEncoder.h:140-145 */
    }
    for (int i = 0; i < 8; ++i) { rc_put(&r, (uint32_t)(r.low >> 56)); r.low <<= 8; }   /* End(): RangeCoder.h:73-80 */
    free(stats);
    return r.pos;
}

/* BinaryRleEncoder: bits[i] != 0 is a "match" (PutSymbol(true)) */
size_t fso_rle_binary(const uint8_t* bits, size_t n, uint8_t* out, size_t cap)
{
    const uint32_t RleMax = 255, RleOffset = 2;
    size_t pos = 0; uint32_t cur = 0;
#define PUT(b) do { if (pos < cap) out[pos] = (uint8_t)(b); pos++; } while (0)
    for (size_t i = 0; i < n; ++i) {
        if (bits[i]) {
            cur++;
            if (cur == RleMax - RleOffset) { PUT(cur + RleOffset); cur = 0; }
        } else {
            int mism = (cur > 0) && (cur < RleMax - RleOffset);
            if (cur > 0) { PUT(cur + RleOffset); cur = 0; }
            if (!mism) PUT(0);
        }
    }
    if (cur > 0) PUT(cur + RleOffset);                                /* End() */
    return pos;
}

/* Rle0Encoder over 32-bit symbols */
size_t fso_rle0(const uint32_t* syms, size_t n, uint8_t* out, size_t cap)
{
    size_t pos = 0; uint32_t prev = 0 /* Rle0BSymbol */;
    for (size_t i = 0; i < n; ++i) {
        uint32_t s = syms[i];
        if (s == 0) {
            if (prev == 0) prev = 1;
            else if (prev == 1) { PUT(0); prev = 0; }
        } else {
            if (prev == 1) { PUT(1); prev = 0; }
            uint32_t ss = s + 1;
            if (ss < 253) PUT(ss);
            else if (ss < (1u << 16) - 1) { PUT(0xFE); PUT(ss >> 8); PUT(ss & 0xFF); }
            else { PUT(0xFF); PUT(ss >> 24); PUT((ss >> 16) & 0xFF); PUT((ss >> 8) & 0xFF); PUT(ss & 0xFF); }
        }
    }
    if (prev == 1) PUT(1);
#undef PUT
    return pos;
}
