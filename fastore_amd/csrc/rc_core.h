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

#if FS_WIDE
// ---- windowed form for the small alphabets (2 and 8 symbols): 64 consecutive symbols of ONE stream per step ----
// Why.  The loop above codes one symbol per trip and every trip begins with a load of the symbol's model -- a random 16-byte
// row of a table of up to 32 MiB, a miss all the way to HBM (with thousands of such tables in flight, a TLB miss too):
// measured 3.5 us per symbol, 25 s for the 7 M-symbol quality stream of one --reduced bin (profiles/r03_reduced_mode.txt).
// But the encoder knows its whole input, and a model's index is a pure function of the input (the ORDER symbols before
// the position and its ctx0, rc/ContextEncoder.h:84-206): it does not depend on what the coder has done.  So the 64 lanes
// take 64 consecutive positions: every lane forms its index and fetches its row -- 64 misses overlap --, positions that
// share a row are told apart by their rank among them, and while a row is not rescaled its statistics at position p are
// those fetched plus 8 for every earlier position of the window on the same row, per symbol (TSymbolCoderRC::EncodeSymbol,
// rc/SymbolCoderRC.h:19-93: +8 on the coded symbol, nothing else) -- three population counts give (frequency, cumulative
// frequency, total).  The last position of every row writes it back.  Then the range coder (RangeCoder.h:40-84) takes the
// triples in stream order, scalar code with one reciprocal per position made by all lanes at once.
// A position whose row reaches the rescale limit ends the window in front of it and is coded by the one-symbol step.
// The bytes are those of the loop above by construction: a different schedule of the same updates.
// q (the two-wave kernel with the windowed coders, fs_encode_streams2_w, since round 4): the triples go to the coder wave
// through the PPMd walk's ring (ppmd_core.h: coder_wave<true>) instead of being coded here -- the model side of window k + 1
// then runs beside the range coder's pass over window k; the stream's size is written by the coder wave, the return value
// is 0.  Measured on a lone 7 M-symbol <8,6> stream: 0.110 -> 0.084 us per symbol (profiles/r04_rc_on_coder_wave.txt).
struct RcQueue { fsppmd::Coder m; FS_GLOBAL uint32_t* sizeOut; uint32_t prio = 0; };
template <int BITS, int ORDER, bool ADV, int CTXBITS>
FS_DEV uint32_t encode_stream_windowed(fs_gptr table, fs_cgptr pairs, uint32_t n, fs_gptr out, uint32_t cap, RcQueue* q = nullptr)
{
    static_assert(BITS == 1 || BITS == 3, "small alphabets only");
    constexpr uint32_t A = 1u << BITS;
    constexpr uint64_t symMask = (1ULL << (ORDER * BITS)) - 1;
    constexpr uint64_t nModels = 1ULL << (BITS * ORDER + (ADV ? CTXBITS : 0));
    constexpr uint32_t limit = (1u << 16) - A * 8u;
    const uint32_t lane = (uint32_t)FS_LANE();
    {   // Clear(): every statistic = 1
        const uint64_t words = nModels * A / 2;
        fs_gptr32 t32 = (fs_gptr32)table;
        if (words >= 4u * FS_WAVE) {
            struct alignas(16) V4 { uint32_t a, b, c, d; };
            FS_GLOBAL V4* t128 = (FS_GLOBAL V4*)table; const V4 ones = {0x00010001u, 0x00010001u, 0x00010001u, 0x00010001u};
            for (uint64_t i = (uint64_t)lane; i < words / 4; i += FS_WAVE) t128[i] = ones;
        } else {
            for (uint64_t i = (uint64_t)lane; i < words; i += FS_WAVE) t32[i] = 0x00010001u;
        }
        FS_WAVE_SYNC();
    }
    Enc e; e.low = 0; e.range = 0xffffffffu; e.out = out; e.cap = cap; e.pos = 0;
    if (q) {      // hand the coder wave this stream's output buffer (the mailbox protocol of fsppmd::encode_member)
        fsppmd::Coder& m = q->m;
        const uint32_t s = FS_UNI(FS_LDS_RD(m.sh->qOpened));
        if (s >= 2u) fsppmd::cq_wait_starts(m, s - 1u);
        if (FS_LANE() == 0) {
            const uint64_t o = (uint64_t)(uintptr_t)out, z = (uint64_t)(uintptr_t)q->sizeOut;
            FS_LDS uint32_t* box = m.sh->qBox[s & 1u];
            box[0] = (uint32_t)o; box[1] = (uint32_t)(o >> 32); box[2] = cap; box[3] = (uint32_t)z; box[4] = (uint32_t)(z >> 32); box[5] = q->prio;
            m.sh->qOpened = s + 1u;
        }
        FS_WAVE_SYNC();
        fsppmd::cq_push(m, fsppmd::CQ_CMD, fsppmd::CQ_START_RC);
    }
    uint64_t hash = 0;                                   // the symbols in front of position k, the latest in the lowest bits
    fs_cgptr16 pairs16 = (fs_cgptr16)pairs;
    for (uint32_t k = 0; k < n;) {
        const uint32_t left = FS_UNI(n - k), W = left < 64u ? left : 64u;
        const bool valid = lane < W;
        const uint32_t pr = valid ? (uint32_t)pairs16[k + lane] : 0u;
        FS_EMU_MEET();
        const uint32_t sym = pr & 0xFFu, ctx = pr >> 8;
        if (fs_ballot(valid && (sym >= A || (ADV && CTXBITS < 8 && ctx >= (1u << CTXBITS)))) != 0ull) {      // (as the loop above: the stream is given up)
            if (q) fsppmd::cq_push(q->m, fsppmd::CQ_CMD, fsppmd::CQ_END_RC_BAD);
            return 0xFFFFFFFFu;
        }
        // the ORDER symbols in front of every position: from the lanes below, and from `hash` for the first lanes
        uint32_t hs = 0;
        #pragma unroll
        for (uint32_t d = 1; d <= (uint32_t)ORDER; ++d) {
            const uint32_t fromLane = fs_bperm(sym, (lane - d) & 63u);
            const uint32_t back = lane < d ? d - lane - 1u : 0u;                                     // d > lane: the (d - lane)-th symbol before k
            const uint32_t fromHash = (uint32_t)(hash >> (BITS * back)) & (A - 1u);
            hs |= (lane >= d ? fromLane : fromHash) << (BITS * (d - 1u));
        }
        const uint32_t h = ADV ? (uint32_t)((((uint64_t)hs & symMask) << CTXBITS) | ctx) : (uint32_t)((uint64_t)hs & symMask);
        fs_gptr16 st = (fs_gptr16)table + (uint64_t)h * A;
        uint32_t v[A];
        if constexpr (A == 8) {
            struct alignas(16) V4 { uint32_t a, b, c, d; };
            V4 r = {0, 0, 0, 0};
            if (valid) r = *(const FS_GLOBAL V4*)st;
            v[0] = r.a & 0xFFFFu; v[1] = r.a >> 16; v[2] = r.b & 0xFFFFu; v[3] = r.b >> 16; v[4] = r.c & 0xFFFFu; v[5] = r.c >> 16; v[6] = r.d & 0xFFFFu; v[7] = r.d >> 16;
        } else {
            uint32_t r = 0;
            if (valid) r = *(const FS_GLOBAL uint32_t*)st;
            v[0] = r & 0xFFFFu; v[1] = r >> 16;
        }
        FS_EMU_MEET();
        // the positions of my row
        uint64_t grp = 0;
        for (uint64_t todo = fs_ballot(valid); todo != 0ull;) {
            const uint32_t j = fs_ctz64(todo), hv = fs_readlane(h, j);
            const uint64_t m = fs_ballot(valid && h == hv);
            if (valid && h == hv) grp = m;
            todo &= ~m;
        }
        const uint64_t b0 = fs_ballot(valid && (sym & 1u) != 0u), b1 = BITS > 1 ? fs_ballot(valid && (sym & 2u) != 0u) : 0ull, b2 = BITS > 2 ? fs_ballot(valid && (sym & 4u) != 0u) : 0ull;
        const uint64_t e0 = (sym & 1u) ? b0 : ~b0, e1 = (sym & 2u) ? b1 : ~b1, e2 = (sym & 4u) ? b2 : ~b2;
        const uint64_t eqS = e0 & e1 & e2;
        const uint64_t ltS = ((sym & 4u) ? ~b2 : 0ull) | (e2 & (((sym & 2u) ? ~b1 : 0ull) | (e1 & ((sym & 1u) ? ~b0 : 0ull))));
        const uint64_t earlier = grp & ((1ull << lane) - 1ull);
        uint32_t tot = 0, below = 0, mine = 0;
        #pragma unroll
        for (uint32_t j = 0; j < A; ++j) { tot += v[j]; below += j < sym ? v[j] : 0u; mine = j == sym ? v[j] : mine; }
        const uint32_t acc = tot + 8u * fs_popc64(earlier), f = mine + 8u * fs_popc64(earlier & eqS), lo = below + 8u * fs_popc64(earlier & ltS);
        // the window ends in front of the first position whose row is due for a rescale
        const uint64_t due = fs_ballot(valid && acc >= limit);
        const uint32_t cnt = FS_UNI(due ? fs_ctz64(due) : W);
        if (cnt != 0u) {
            const uint64_t inWin = cnt >= 64u ? ~0ull : (1ull << cnt) - 1ull;
            const uint64_t g = grp & inWin;
            // the last position of a row writes it back: what was fetched + 8 per position of the row, symbol by symbol
            const bool writer = lane < cnt && (g >> lane) == 1ull;
            uint32_t nv[A];
            #pragma unroll
            for (uint32_t j = 0; j < A; ++j) {
                const uint64_t isJ = ((j & 1u) ? b0 : ~b0) & ((j & 2u) ? b1 : ~b1) & ((j & 4u) ? b2 : ~b2);
                nv[j] = v[j] + 8u * fs_popc64(g & isJ);
            }
            if constexpr (A == 8) {
                struct alignas(16) V4 { uint32_t a, b, c, d; };
                if (writer) { const V4 w = {nv[0] | (nv[1] << 16), nv[2] | (nv[3] << 16), nv[4] | (nv[5] << 16), nv[6] | (nv[7] << 16)}; *(FS_GLOBAL V4*)st = w; }
            } else {
                if (writer) *(FS_GLOBAL uint32_t*)st = nv[0] | (nv[1] << 16);
            }
            FS_EMU_MEET();
            if (q) fsppmd::cq_push_lanes(q->m, lo | ((f & 0x3FFFu) << 16) | fsppmd::CQ_RC, acc | ((f >> 14) << 16), cnt);
            else
            {
            // the range coder over the triples, in stream order; range / total by a reciprocal per position (total in [2, 65535])
            uint32_t rmul = 0, rl = 1;
            if (lane < cnt) { const fsppmd::Recip rc = fsppmd::recip_make(acc); rmul = rc.mul; rl = rc.l; }
            // (the coder's state is wave-uniform by construction; said so, the pass is scalar code with scalar branches -- left to the
            // compiler's divergence analysis it was vector code under exec masks, two saved masks per symbol and two per byte)
            uint64_t low = ((uint64_t)FS_UNI((uint32_t)(e.low >> 32)) << 32) | FS_UNI((uint32_t)e.low);
            uint32_t range = FS_UNI(e.range), pos = FS_UNI(e.pos);
            const uint32_t capU = FS_UNI(e.cap);
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint32_t F = fs_readlane(f, i), LO = fs_readlane(lo, i), M = fs_readlane(rmul, i), L = fs_readlane(rl, i);
                range = fsppmd::recip_div(range, M, L);
                low += (uint32_t)(range * LO);
                range *= F;
                while (range <= 0x00ffffffu) {
                    if ((low ^ (low + range)) & 0xff00000000000000ULL) { const uint32_t x = (uint32_t)low; range = (x | 0x00ffffffu) - x; }
                    if (pos < capU) fs_st8(e.out + pos, (uint32_t)(low >> 56));
                    pos++;
                    low <<= 8; range <<= 8;
                }
            }
            e.low = low; e.range = range; e.pos = pos;
            }
            // the symbols in front of position k + cnt
            uint64_t nh = 0;
            #pragma unroll
            for (uint32_t d = 1; d <= (uint32_t)ORDER; ++d) {
                const uint32_t s = cnt >= d ? fs_readlane(sym, cnt - d) : (uint32_t)(hash >> (BITS * (cnt < d ? d - cnt - 1u : 0u))) & (A - 1u);
                nh |= (uint64_t)s << (BITS * (d - 1u));
            }
            hash = nh;
            k += cnt;
        }
        if (cnt < W) {      // position k: its row is due for a rescale -- the one-symbol step of the loop above
            const uint32_t pr1 = fs_ld16(pairs + 2u * k);
            const uint32_t s1 = pr1 & 0xFFu, c1 = pr1 >> 8;
            const uint32_t h1 = ADV ? (uint32_t)(((hash & symMask) << CTXBITS) | c1) : (uint32_t)(hash & symMask);
            fs_gptr16 st1 = (fs_gptr16)table + (uint64_t)h1 * A;
            uint32_t w[A], acc1 = 0;
            const uint32_t own = lane < A ? (uint32_t)st1[lane] : 0u;
            FS_EMU_MEET();
            for (uint32_t j = 0; j < A; ++j) { w[j] = fs_readlane(own, j); acc1 += w[j]; }
            if (acc1 >= limit) {
                acc1 = 0;
                for (uint32_t j = 0; j < A; ++j) { w[j] -= w[j] >> 1; acc1 += w[j]; }
                if (lane < A) st1[lane] = (uint16_t)(own - (own >> 1));
                FS_EMU_MEET();
            }
            uint32_t lo1 = 0, f1 = 0;
            for (uint32_t j = 0; j < A; ++j) { if (j < s1) lo1 += w[j]; if (j == s1) f1 = w[j]; }
            fs_st16((fs_gptr)(st1 + s1), f1 + 8);
            FS_EMU_MEET();
            if (q) fsppmd::cq_push(q->m, lo1 | ((f1 & 0x3FFFu) << 16) | fsppmd::CQ_RC, acc1 | ((f1 >> 14) << 16));
            else
            encode_freq(e, f1, lo1, acc1);
            hash = (hash << BITS) | s1;
            ++k;
        }
    }
    if (q) { fsppmd::cq_push(q->m, fsppmd::CQ_CMD, fsppmd::CQ_END_RC); return 0u; }
    for (int i = 0; i < 8; ++i) { put(e, (uint32_t)(e.low >> 56)); e.low <<= 8; }
    return e.pos;
}
#endif

// model ids shared by host and device
enum Model : uint32_t { M_S2O4 = 0, M_S8O4 = 1, M_A8O4 = 2, M_A2O10 = 3, M_A8O6 = 4, M_A256O1 = 5, M_A256O1_C6 = 6 /* ctx0 < 64: 8 MiB table */, M_COUNT = 7 };

// the one-symbol loop for every model (the kernels of launches without a long range-coded stream: their code is what it was)
FS_DEV uint32_t encode_model_serial(uint32_t model, fs_gptr table, fs_cgptr pairs, uint32_t n, fs_gptr out, uint32_t cap)
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
FS_DEV uint32_t encode_model(uint32_t model, fs_gptr table, fs_cgptr pairs, uint32_t n, fs_gptr out, uint32_t cap)
{
#if FS_WIDE && !defined(FS_RC_SERIAL)
    switch (model) {      // the small alphabets: 64 symbols per step
    case M_S2O4: return encode_stream_windowed<1, 4, false, 1>(table, pairs, n, out, cap);
    case M_S8O4: return encode_stream_windowed<3, 4, false, 3>(table, pairs, n, out, cap);
    case M_A8O4: return encode_stream_windowed<3, 4, true, 3>(table, pairs, n, out, cap);
    case M_A2O10: return encode_stream_windowed<1, 10, true, 1>(table, pairs, n, out, cap);
    case M_A8O6: return encode_stream_windowed<3, 6, true, 3>(table, pairs, n, out, cap);
    default: break;
    }
#endif
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
#if FS_WIDE
// the small alphabets with their triples sent to the coder wave; false: not a model with a windowed form (the caller codes it itself)
FS_DEV bool encode_model_queued(uint32_t model, fs_gptr table, fs_cgptr pairs, uint32_t n, fs_gptr out, uint32_t cap, RcQueue* q)
{
    switch (model) {
    case M_S2O4: (void)encode_stream_windowed<1, 4, false, 1>(table, pairs, n, out, cap, q); return true;
    case M_S8O4: (void)encode_stream_windowed<3, 4, false, 3>(table, pairs, n, out, cap, q); return true;
    case M_A8O4: (void)encode_stream_windowed<3, 4, true, 3>(table, pairs, n, out, cap, q); return true;
    case M_A2O10: (void)encode_stream_windowed<1, 10, true, 1>(table, pairs, n, out, cap, q); return true;
    case M_A8O6: (void)encode_stream_windowed<3, 6, true, 3>(table, pairs, n, out, cap, q); return true;
    default: return false;
    }
}
#endif
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
