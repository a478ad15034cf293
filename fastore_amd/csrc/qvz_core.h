// QVZ adaptive arithmetic coder: one quality stream per wavefront (device) or per call (host build).
//
// Restates the reference's coder for the quantizer-state symbols of --lossy archives:
//   arithmetic_encoder_step / encoder_last_step   fastore_pack/arith.cpp:33-125 (m = 22 bit registers, E1/E2/E3 rescaling)
//   update_stats / initialize_stream_stats        fastore_pack/qv_stream.cpp:19-71 (step 8, halve-plus-one past 2^19)
//   QVZEncoder::Start / EncodeNext / End          fastore_pack/qv_compressor.h:139-160
//   bit output                                    fastore_bin/BitMemory.h:251-313, 392-404 (MSB first, zero padded to a byte)
//
// Input: one u32 per quality value, context | state << 24, produced by the front end (qvz.cpp), where a
// context is one conditional quantizer (column, previous quantized value, low/high).  The model blob is
// read-only and shared by every stream of the library:
//     ModelHeader | Desc[n_ctx] | image[image_words]
// `image` is the initial statistics of all contexts; each context owns image[off .. off + card]:
// word 0 = n (the total), words 1..card = the counts.  A stream copies the image into its private arena
// and adapts it there.  Lane j of the wave always touches word j of a context block, so a wave's own
// program order is all the ordering the updates need.
#pragma once
#include "wave.h"
#include "ppmd_core.h"      // (the two-wave kernels: a stream's symbols through the PPMd walk's ring to the coder wave)

namespace fsqvz {

enum : uint32_t { M_BITS = 22, STEP = 8, RESCALE_AT = 1u << (M_BITS - 3), MAX_CARD = 72 };

struct ModelHeader { uint32_t n_ctx, image_words, columns, reserved; };
struct Desc { uint32_t off, card; };

FS_DEV uint32_t blob_bytes(uint32_t n_ctx, uint32_t image_words) { return (uint32_t)sizeof(ModelHeader) + n_ctx * (uint32_t)sizeof(Desc) + image_words * 4u; }

struct BitOut {
    fs_gptr out; uint32_t pos, cap; uint64_t acc; uint32_t nb; uint32_t overflow;
};
FS_DEV void put_bits(BitOut& o, uint32_t v, uint32_t k)        // k <= 32, fewer than 8 bits pending on entry
{
    o.acc = (o.acc << k) | v; o.nb += k;
    while (o.nb >= 8) {
        o.nb -= 8;
        if (o.pos < o.cap) o.out[o.pos] = (uint8_t)(o.acc >> o.nb); else o.overflow = 1;
        o.pos++;
    }
}
// one decided bit followed by the pending opposite bits (arith.cpp:80-92)
FS_DEV void put_decided(BitOut& o, uint32_t bit, uint32_t& scale3)
{
    put_bits(o, bit, 1);
    const uint32_t inv = bit ^ 1u;
    while (scale3 >= 24) { put_bits(o, inv ? 0xFFFFFFu : 0u, 24); scale3 -= 24; }
    if (scale3) { put_bits(o, inv ? ((1u << scale3) - 1u) : 0u, scale3); scale3 = 0; }
}

// cumulative count below x, the count of x and the total of the context block `blk`; then the adaptive update
FS_DEV void model_step(FS_GLOBAL uint32_t* blk, uint32_t card, uint32_t x, uint32_t& cumLo, uint32_t& cnt, uint32_t& total)
{
#if FS_WIDE
    const uint32_t lane = (uint32_t)FS_LANE();
    // words 0..card of the block: lane j holds word j, lanes 0..8 also hold word 64 + j (card <= 72)
    uint32_t v = lane <= card ? blk[lane] : 0u;
    uint32_t part = (lane >= 1u && lane <= x) ? v : 0u;
    uint32_t v2 = 0;
    if (card >= 64u) {                                   // uniform
        v2 = 64u + lane <= card ? blk[64u + lane] : 0u;
        part += (64u + lane <= x) ? v2 : 0u;
    }
    FS_EMU_MEET();
    // (the sum over the lanes by DPP row shifts and broadcasts -- six adds; as six __shfl_xor steps it was six trips through the
    // LDS crossbar, one behind the other, in every symbol's chain)
    cumLo = fs_wave_sum8(part, true);
    total = fs_readlane(v, 0);
    const uint32_t w = x + 1u;                           // word of the coded symbol
    cnt = w < 64u ? fs_readlane(v, w) : fs_readlane(v2, w - 64u);
    uint32_t nt = total + STEP;
    if (nt > RESCALE_AT) {                               // uniform, once per 65536 uses of a context at most
        uint32_t c = (lane == w) ? v + STEP : v;
        uint32_t c2 = (64u + lane == w) ? v2 + STEP : v2;
        if (lane >= 1u && lane <= card && c) c = (c >> 1) + 1u;
        if (64u + lane <= card && c2) c2 = (c2 >> 1) + 1u;
        uint32_t s = ((lane >= 1u && lane <= card) ? c : 0u) + ((64u + lane <= card) ? c2 : 0u);
        nt = fs_wave_sum8(s, true);
        if (lane >= 1u && lane <= card) blk[lane] = c;
        if (64u + lane <= card) blk[64u + lane] = c2;
        if (lane == 0u) blk[0] = nt;
    } else {
        if (lane == 0u) blk[0] = nt;
        if (lane == w) blk[w] = v + STEP;
        if (64u + lane == w) blk[w] = v2 + STEP;
    }
    FS_EMU_MEET();
#else
    uint32_t lo = 0;
    for (uint32_t i = 0; i < x; ++i) lo += blk[1 + i];
    cumLo = lo; cnt = blk[1 + x]; total = blk[0];
    blk[1 + x] += STEP; blk[0] += STEP;
    if (blk[0] > RESCALE_AT) {
        uint32_t n = 0;
        for (uint32_t i = 0; i < card; ++i) if (blk[1 + i]) { blk[1 + i] = (blk[1 + i] >> 1) + 1u; n += blk[1 + i]; }
        blk[0] = n;
    }
#endif
}

// returns the stream size in bytes, or 0xFFFFFFFF when `cap` is too small or a symbol is malformed
FS_DEV uint32_t encode_stream(fs_gptr arena, fs_cgptr model, fs_cgptr in, uint32_t n, fs_gptr out, uint32_t cap)
{
    const FS_GLOBAL ModelHeader* hdr = (const FS_GLOBAL ModelHeader*)model;
    const uint32_t nCtx = FS_UNI(hdr->n_ctx), words = FS_UNI(hdr->image_words);
    const FS_GLOBAL Desc* desc = (const FS_GLOBAL Desc*)(model + sizeof(ModelHeader));
    fs_cgptr image = model + sizeof(ModelHeader) + (uint64_t)nCtx * sizeof(Desc);
    fs_wave_copy4(arena, image, words * 4u);
    FS_GLOBAL uint32_t* stat = (FS_GLOBAL uint32_t*)arena;

    BitOut o; o.out = out; o.pos = 0; o.cap = cap; o.acc = 0; o.nb = 0; o.overflow = 0;
    uint32_t l = 0, u = (1u << M_BITS) - 1u, scale3 = 0, bad = 0;
    const uint32_t msbShift = M_BITS - 1, smsbShift = M_BITS - 2, clearMask = (1u << msbShift) - 1u;
    const FS_GLOBAL uint32_t* sym = (const FS_GLOBAL uint32_t*)in;

    // The symbol word and its context's descriptor do not depend on the coder: those of symbol i + 1 are requested while
    // symbol i is coded (two of the three dependent memory trips of a symbol, out of its chain).
    uint32_t wNext = n ? sym[0] : 0u, offNext = 0, cardNext = 0;
    if (n && (wNext & 0xFFFFFFu) < nCtx) { offNext = desc[wNext & 0xFFFFFFu].off; cardNext = desc[wNext & 0xFFFFFFu].card; }
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t w = FS_UNI(wNext);
        const uint32_t ctx = w & 0xFFFFFFu, x = w >> 24;
        if (ctx >= nCtx) { bad = 1; break; }
        const uint32_t off = FS_UNI(offNext), card = FS_UNI(cardNext);
        if (i + 1u < n) {
            wNext = sym[i + 1u];
            const uint32_t cN = wNext & 0xFFFFFFu;
            if (cN < nCtx) { offNext = desc[cN].off; cardNext = desc[cN].card; }
        }
        if (x >= card || card > MAX_CARD) { bad = 1; break; }
        uint32_t cumLo, cnt, total;
        model_step(stat + off, card, x, cumLo, cnt, total);
        // (range <= 2^22, counts <= total < 2^20: the products are below 2^42 and the quotients below 2^22; a double holds both
        // operands exactly and its quotient is off by less than 2^-31, while a quotient that is not a whole number is at least
        // 1 / total > 2^-20 away from the next one: truncation gives the integer quotient of arith.cpp:44-45, without the
        // hundred-instruction 64-bit division)
        const uint64_t range = (uint64_t)u - l + 1u;
        u = l + (uint32_t)((double)(range * (cumLo + cnt)) / (double)total) - 1u;
        l = l + (uint32_t)((double)(range * cumLo) / (double)total);
        for (;;) {
            const uint32_t msbL = l >> msbShift, msbU = u >> msbShift;
            if (msbL == msbU) {
                put_decided(o, msbL, scale3);
                l = (l & clearMask) << 1;
                u = ((u & clearMask) << 1) + 1u;
            } else if ((l >> smsbShift) == 1u && (u >> smsbShift) == 2u) {
                scale3 += 1u;
                u = (((u << 1) & clearMask) | (1u << msbShift)) + 1u;
                l = (l << 1) & clearMask;
            } else break;
        }
    }
    // encoder_last_step: the msb of the tag, the pending bits, the other m-1 tag bits, zero padding
    const uint32_t msbL = l >> msbShift;
    put_decided(o, msbL, scale3);
    put_bits(o, l & clearMask, M_BITS - 1);
    if (o.nb) put_bits(o, 0u, 8u - o.nb);
    FS_WAVE_SYNC();
    return (o.overflow || bad) ? 0xFFFFFFFFu : o.pos;
}

#if FS_WIDE
// ---- windowed form (the product's kernels since round 4) ----
// A symbol's context is a pure function of the input (fs_gather_quality_qvz put it into the symbol's word), and the contexts of
// consecutive symbols differ -- a context belongs to a column -- so 64 lanes take 64 consecutive symbols: every lane fetches
// its descriptor, then its context's total and its counts up to its symbol in one go; no wave-wide sum, no chain of three
// dependent loads per symbol.  A window ends in front of the first symbol that shares its context with an earlier one of the
// window, whose context is due for a rescale, or that is malformed; that symbol takes the one-symbol step.  The arithmetic coder
// (arith.cpp:33-125) then passes over the window's (low count, count, total) triples in stream order -- with the reciprocals of
// all the window's totals made beforehand, one per lane, and the interval's rescaling in closed form (below).
//
// Shared contexts.  Contexts are numbered by column, so inside one read the context numbers of a window rise; a read boundary
// shows as a descent.  No descent: no two symbols share a context.  One descent at lane b (the usual case when a window holds
// the end of one read and the start of the next): a symbol of the second read can only share with the first read's part if its
// context number reaches that of lane 0, the smallest there.  Anything else (reads shorter than a window) is compared pair by pair.
// floor(range * c / total) for range <= 2^22, c <= total < 2^20 (arith.cpp:44-45 computes it by a 64-bit integer division), with
// the fraction c / total made beforehand -- by all the window's lanes at once -- as F = ceil(c * 2^42 / total), exactly: then
// (range * F) >> 42 IS the quotient.  (F exceeds the true fraction by less than 2^-42, range times that is below 2^-20 < 1 / total,
// the least distance of a quotient that is not whole from the next whole number; a whole quotient is met from above.)  The product
// needs 64 bits, its bits from 42 on only mulhi(range, F's low word) + range * (F's high word): three scalar multiplications and
// no comparison in the symbol's chain.
struct QFrac { uint32_t lo, hi; };
FS_DEV QFrac q_frac(uint32_t c, uint32_t total)
{
    // an estimate from the double-precision quotient (good to a few units), set right by the remainder: n = c * 2^42 = e * total + r
    const uint64_t n = (uint64_t)c << 42;
    uint64_t e = (uint64_t)((double)c * 4398046511104.0 / (double)total);
    int64_t r = (int64_t)(n - e * total);
    while (r < 0) { --e; r += total; }
    while (r >= (int64_t)total) { ++e; r -= total; }
    e += r != 0 ? 1u : 0u;                                               // ceil
    QFrac f; f.lo = (uint32_t)e; f.hi = (uint32_t)(e >> 32);
    return f;
}
FS_DEV uint32_t q_mulhi(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
// (c == total: F = 2^42, whose high word times a full range leaves 32 bits -- the quotient is the range itself)
FS_DEV uint32_t q_div(uint32_t range, uint32_t fLo, uint32_t fHi) { const uint32_t q = (q_mulhi(range, fLo) + range * fHi) >> 10; return (fHi >> 10) ? range : q; }
// One symbol through the interval: arithmetic_encoder_step (arith.cpp:33-103), the E1/E2/E3 loop in closed form.  The loop
// first shifts out the leading bits that l and u share (each is a decided bit; the pending opposite bits follow the first),
// and only then -- the top bits now differ: l = 0.., u = 1.. -- counts the steps in which l continues 01 and u continues 10
// (E3: the top bits stay different, so no decided bit can follow).  So: k1 = the length of the common prefix of l and u, k3 = the
// length of the run, behind the top bit, of positions where l has a one and u a zero.
FS_DEV void q_code(BitOut& o, uint32_t& l, uint32_t& u, uint32_t& scale3, QFrac below, QFrac upto)
{
    const uint32_t M22 = (1u << M_BITS) - 1u, clearMask = (1u << (M_BITS - 1)) - 1u;
    const uint32_t range = u - l + 1u;                       // <= 2^22
    u = l + q_div(range, upto.lo, upto.hi) - 1u;
    l = l + q_div(range, below.lo, below.hi);
    const uint32_t x = (l ^ u) & M22;
    const uint32_t k1 = x ? (uint32_t)__builtin_clz(x) - (32u - M_BITS) : (uint32_t)M_BITS;
    if (k1) {
        const uint32_t top = l >> (M_BITS - k1);             // the k1 shared bits, first decided bit on top
        put_decided(o, top >> (k1 - 1u), scale3);
        if (k1 > 1u) put_bits(o, top & ((1u << (k1 - 1u)) - 1u), k1 - 1u);
        l = (l << k1) & M22; u = ((u << k1) | ((1u << k1) - 1u)) & M22;
    }
    // (here l = 0.., u = 1.. -- or k1 = 22 and l = 0, u = 2^22 - 1, the same shape)
    const uint32_t t = (l << 1) & (~u << 1) & M22;            // bit 21 - j: the step j + 1 of the run would be an E3 step
    const uint32_t nt = ~t & M22;
    const uint32_t k3 = nt ? (uint32_t)__builtin_clz(nt) - (32u - M_BITS) : (uint32_t)M_BITS - 1u;
    if (k3) {
        scale3 += k3;
        l = (l << k3) & clearMask;
        u = ((u << k3) & clearMask) | (1u << (M_BITS - 1)) | ((1u << k3) - 1u);
    }
}

// The coder wave's side of a QVZ stream in the two-wave kernels (fsppmd::coder_wave<true, WaveCoder>): a batch's entries become fractions --
// all lanes at once --, then the interval passes over them in stream order.
struct WaveCoder {
    static constexpr bool on = true;
    struct State { BitOut o; uint32_t l, u, scale3; };
    FS_DEV_M static void start(State& s, fs_gptr out, uint32_t cap) { s.o.out = out; s.o.pos = 0; s.o.cap = cap; s.o.acc = 0; s.o.nb = 0; s.o.overflow = 0; s.l = 0; s.u = (1u << M_BITS) - 1u; s.scale3 = 0; }
    FS_DEV_M static void prepare(uint32_t A, uint32_t M, uint32_t& bLo, uint32_t& bHi, uint32_t& uLo, uint32_t& uHi)
    {
        const uint32_t cumLo = A & 0xFFFFFu, cnt = ((A >> 20) & 0x3FFu) | (((M >> 20) & 0x3FFu) << 10), total = M & 0xFFFFFu;
        const QFrac fB = q_frac(cumLo, total ? total : 1u), fU = q_frac(cumLo + cnt, total ? total : 1u);
        bLo = fB.lo; bHi = fB.hi; uLo = fU.lo; uHi = fU.hi;
    }
    FS_DEV_M static void code(State& s, uint32_t bLo, uint32_t bHi, uint32_t uLo, uint32_t uHi)
    {
        QFrac b, u; b.lo = bLo; b.hi = bHi; u.lo = uLo; u.hi = uHi;
        s.l = FS_UNI(s.l); s.u = FS_UNI(s.u); s.scale3 = FS_UNI(s.scale3);
        q_code(s.o, s.l, s.u, s.scale3, b, u);
    }
    FS_DEV_M static uint32_t finish(State& s)      // encoder_last_step: the msb of the tag, the pending bits, the other m-1 tag bits, zero padding
    {
        const uint32_t msbShift = M_BITS - 1, clearMask = (1u << msbShift) - 1u;
        put_decided(s.o, s.l >> msbShift, s.scale3);
        put_bits(s.o, s.l & clearMask, M_BITS - 1);
        if (s.o.nb) put_bits(s.o, 0u, 8u - s.o.nb);
        FS_WAVE_SYNC();
        return s.o.overflow ? 0xFFFFFFFFu : s.o.pos;
    }
};
// q (the two-wave kernel fs_encode_streams2_w): the symbols' counts go to the coder wave through the PPMd walk's ring instead of being coded
// here -- fetching and updating the contexts of window k + 1 then runs beside the fractions and the interval's pass over window k; the
// stream's size is written by the coder wave, the return value is 0
struct QvzQueue { fsppmd::Coder m; FS_GLOBAL uint32_t* sizeOut; uint32_t prio = 0; };
FS_DEV uint32_t encode_stream_windowed(fs_gptr arena, fs_cgptr model, fs_cgptr in, uint32_t n, fs_gptr out, uint32_t cap, QvzQueue* q = nullptr)
{
    const FS_GLOBAL ModelHeader* hdr = (const FS_GLOBAL ModelHeader*)model;
    const uint32_t nCtx = FS_UNI(hdr->n_ctx), words = FS_UNI(hdr->image_words);
    const FS_GLOBAL Desc* desc = (const FS_GLOBAL Desc*)(model + sizeof(ModelHeader));
    fs_cgptr image = model + sizeof(ModelHeader) + (uint64_t)nCtx * sizeof(Desc);
    fs_wave_copy4(arena, image, words * 4u);
    FS_GLOBAL uint32_t* stat = (FS_GLOBAL uint32_t*)arena;
    const uint32_t lane = (uint32_t)FS_LANE();

    BitOut o; o.out = out; o.pos = 0; o.cap = cap; o.acc = 0; o.nb = 0; o.overflow = 0;
    uint32_t l = 0, u = (1u << M_BITS) - 1u, scale3 = 0, bad = 0;
    const uint32_t msbShift = M_BITS - 1, clearMask = (1u << msbShift) - 1u;
    const FS_GLOBAL uint32_t* sym = (const FS_GLOBAL uint32_t*)in;
    if (q) {      // hand the coder wave this stream's output buffer (the mailbox protocol of fsppmd::encode_member)
        fsppmd::Coder& m = q->m;
        const uint32_t s = FS_UNI(FS_LDS_RD(m.sh->qOpened));
        if (s >= 2u) fsppmd::cq_wait_starts(m, s - 1u);
        if (FS_LANE() == 0) {
            const uint64_t o64 = (uint64_t)(uintptr_t)out, z = (uint64_t)(uintptr_t)q->sizeOut;
            FS_LDS uint32_t* box = m.sh->qBox[s & 1u];
            box[0] = (uint32_t)o64; box[1] = (uint32_t)(o64 >> 32); box[2] = cap; box[3] = (uint32_t)z; box[4] = (uint32_t)(z >> 32); box[5] = q->prio;
            m.sh->qOpened = s + 1u;
        }
        FS_WAVE_SYNC();
        fsppmd::cq_push(m, fsppmd::CQ_CMD, fsppmd::CQ_START_QVZ);
    }
    for (uint32_t k = 0; k < n && !bad;) {
        const uint32_t left = FS_UNI(n - k), W = left < 64u ? left : 64u;
        const bool valid = lane < W;
        const uint32_t w = valid ? sym[k + lane] : 0u;
        const uint32_t ctx = w & 0xFFFFFFu, x = w >> 24;
        bool wrong = valid && ctx >= nCtx;
        uint32_t off = 0, card = 0;
        if (valid && !wrong) { off = desc[ctx].off; card = desc[ctx].card; }
        wrong = wrong || (valid && (x >= card || card > MAX_CARD));
        FS_EMU_MEET();
        // an earlier symbol of the window in the same context?
        bool shared = false;
        {
            const uint32_t before = fs_bperm(ctx, (lane + 63u) & 63u);
            const uint64_t descents = fs_ballot(valid && lane > 0u && ctx <= before);      // (equal neighbours count: they share)
            bool slow = false;
            if (descents != 0ull) {
                const uint32_t b = fs_ctz64(descents), first = fs_readlane(ctx, 0);
                if ((descents & (descents - 1ull)) != 0ull) slow = true;                   // more than one read boundary in the window
                else if (fs_ballot(valid && lane >= b && ctx >= first) != 0ull) slow = true;
            }
            if (FS_UB(slow))
                for (uint32_t d = 1; d < W; ++d) {
                    const uint32_t other = fs_bperm(ctx, (lane - d) & 63u);
                    shared = shared || (valid && lane >= d && other == ctx);
                }
        }
        // my context's total, my symbol's count, the counts below it: the block's first eight words in one go, the rest in turn
        uint32_t total = 1, cnt = 0, cumLo = 0;
        const bool fetch = valid && !wrong;
        FS_GLOBAL uint32_t* blk = stat + (fetch ? off : 0u);
        {
            uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0;
            if (fetch) { b0 = blk[0]; b1 = blk[1]; if (card >= 2u) b2 = blk[2]; if (card >= 3u) b3 = blk[3]; if (card >= 4u) b4 = blk[4]; if (card >= 5u) b5 = blk[5]; if (card >= 6u) b6 = blk[6]; if (card >= 7u) b7 = blk[7]; }
            total = fetch ? b0 : 1u;
            cumLo = (x > 0u ? b1 : 0u) + (x > 1u ? b2 : 0u) + (x > 2u ? b3 : 0u) + (x > 3u ? b4 : 0u) + (x > 4u ? b5 : 0u) + (x > 5u ? b6 : 0u) + (x > 6u ? b7 : 0u);
            cnt = x == 0u ? b1 : (x == 1u ? b2 : (x == 2u ? b3 : (x == 3u ? b4 : (x == 4u ? b5 : (x == 5u ? b6 : b7)))));
            if (fs_ballot(fetch && x > 6u) != 0ull) {
                if (fetch && x > 6u) cnt = blk[1u + x];
                for (uint32_t j = 7u; fs_ballot(fetch && j < x) != 0ull; ++j) if (fetch && j < x) cumLo += blk[1u + j];
            }
            if (!fetch) { cumLo = 0; cnt = 0; }
        }
        FS_EMU_MEET();
        const bool due = fetch && total + STEP > RESCALE_AT;
        const uint64_t stop = fs_ballot(valid && (wrong || shared || due));
        const uint32_t take = FS_UNI(stop ? fs_ctz64(stop) : W);
        // the window's updates: every context is there once
        if (lane < take) { blk[0] = total + STEP; blk[1u + x] = cnt + STEP; }
        FS_EMU_MEET();
        if (q) {
            if (take) fsppmd::cq_push_lanes(q->m, cumLo | ((cnt & 0x3FFu) << 20), total | ((cnt >> 10) << 20) | fsppmd::CQ_QVZ, take);
        } else {
            // every position's two fractions, all lanes at once
            const uint32_t tsafe = total ? total : 1u;
            const QFrac fB = q_frac(cumLo, tsafe), fU = q_frac(cumLo + cnt, tsafe);
            l = FS_UNI(l); u = FS_UNI(u); scale3 = FS_UNI(scale3);
            for (uint32_t i = 0; i < take; ++i) {
                QFrac b1, u1;
                b1.lo = FS_UNI(fs_readlane(fB.lo, i)); b1.hi = FS_UNI(fs_readlane(fB.hi, i)); u1.lo = FS_UNI(fs_readlane(fU.lo, i)); u1.hi = FS_UNI(fs_readlane(fU.hi, i));
                q_code(o, l, u, scale3, b1, u1);
            }
        }
        k += take;
        if (take < W) {          // symbol k: shares its context, is due for a rescale, or is malformed -- the one-symbol step
            const uint32_t w1 = FS_UNI(fs_readlane(w, take));
            const uint32_t ctx1 = w1 & 0xFFFFFFu, x1 = w1 >> 24;
            if (ctx1 >= nCtx) { bad = 1; break; }
            const uint32_t off1 = FS_UNI(desc[ctx1].off), card1 = FS_UNI(desc[ctx1].card);
            if (x1 >= card1 || card1 > MAX_CARD) { bad = 1; break; }
            uint32_t c1, n1, t1;
            model_step(stat + off1, card1, x1, c1, n1, t1);
            if (q) fsppmd::cq_push(q->m, c1 | ((n1 & 0x3FFu) << 20), t1 | ((n1 >> 10) << 20) | fsppmd::CQ_QVZ);
            else q_code(o, l, u, scale3, q_frac(c1, t1), q_frac(c1 + n1, t1));
            ++k;
        }
    }
    if (q) { fsppmd::cq_push(q->m, fsppmd::CQ_CMD, bad ? fsppmd::CQ_END_QVZ_BAD : fsppmd::CQ_END_QVZ); return 0u; }
    const uint32_t msbL = l >> msbShift;
    put_decided(o, msbL, scale3);
    put_bits(o, l & clearMask, M_BITS - 1);
    if (o.nb) put_bits(o, 0u, 8u - o.nb);
    FS_WAVE_SYNC();
    return (o.overflow || bad) ? 0xFFFFFFFFu : o.pos;
}
#else
struct WaveCoder : fsppmd::NoQvz {};      // (the compiler's host pass over the kernels names it)
#endif

}  // namespace fsqvz
