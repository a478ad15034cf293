// Windowed hit path of the PPMd encoder: up to 64 consecutive symbols of ONE stream per step, one position per lane.
// Included by ppmd_core.h inside namespace fsppmd (64-lane builds only).
//
// Why.  PPMd is a serial state machine, and a wavefront that walks it one symbol at a time spends ~300 instructions and
// two to three dependent memory round trips per symbol.  But the encoder knows its whole input, and on
// predictable data (quality strings: 99 % of the symbols of a long stream) the walk is a run of PLAIN HITS: the symbol
// is found in the current context at full order (OrderFall == 0) and the found state's successor is a real context, so
// the reference neither escapes nor calls UpdateModel (Model.cpp:559-586, the branch at :574).  In such a run
//   * the context of position i is a pure function of the model's (unchanged) tree and the four bytes in front of i;
//   * a plain hit touches nothing but its own context: frequency +4, SummFreq +4, one swap, now and then a rescale
//     (update1 / encodeSymbol1, Model.cpp:447-481; rescale :246-280) -- no allocation, no pointer changes;
//   * the only state that chains through all positions is the range coder (low, range) and PrevSuccess.
// So the 64 lanes take 64 consecutive positions: each lane looks its context up in a HINT table (last four bytes ->
// context index, learned by the serial path), fetches the record and the state list, finds its symbol, and the wave
// checks the chain (successor of position i-1 == context of position i) -- which proves every hint that passes, so the
// table never has to be right, only mostly right.  Positions that share a context are ordered by rank and processed in
// rounds (the state travels from lane to lane by ds_bpermute); the results -- one (cumulative frequency, frequency,
// total) triple per position -- then go through the range coder in stream order, scalar code with a precomputed
// reciprocal per position.  Anything else (escape, binary context, more than eight states, a rescale that frees
// units, a missing or stale hint) ends the window in front of that position; the serial path codes it.
// The bytes are those of the serial walk by construction: the window is only a different schedule of the same updates.
#pragma once

enum : uint32_t { HINT_BITS = 16u, WIN_MAX_NS = 7u /* NumStats field: up to eight states */ };

FS_DEV uint32_t hint_slot(uint32_t key) { return (key * 0x9E3779B1u) >> (32u - HINT_BITS); }
// the serial path saw `ctx` as the full-order context behind the four bytes `key`
FS_DEV void hint_learn(Coder& m, uint32_t key, uint32_t ctx)
{
    fs_gptr32 e = (fs_gptr32)(m.hb + 1u + HINT_OFF + 8u * hint_slot(key));
    e[0] = key; e[1] = ctx;
}

// n / d for d in [2, 65535] by multiplication (Granlund & Montgomery 1994, fig. 4.1, N = 32):
//   l = ceil(log2 d), m' = floor(2^32 (2^l - d) / d) + 1, t = mulhi(m', n), q = (t + ((n - t) >> 1)) >> (l - 1)
struct Recip { uint32_t mul, l; };
FS_DEV Recip recip_make(uint32_t d)
{
    Recip r;
    r.l = 32u - (uint32_t)__builtin_clz(d - 1u);                 // d >= 2
    const uint32_t e = (1u << r.l) - d;                         // < d < 2^16
    const uint32_t hi = (e << 16) / d, rem = (e << 16) - hi * d; // two 32-bit divisions give floor(e * 2^32 / d)
    const uint32_t lo = (rem << 16) / d;
    r.mul = ((hi << 16) | lo) + 1u;
    return r;
}
FS_DEV uint32_t fs_mulhi(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
FS_DEV uint32_t recip_div(uint32_t n, uint32_t mul, uint32_t l) { const uint32_t t = fs_mulhi(n, mul); return (t + ((n - t) >> 1)) >> (l - 1u); }

// compare-exchange of two packed states, larger `hi` first
#define FS_CE(a, b) do { const bool sw_ = hi[a] < hi[b]; const uint32_t h_ = sw_ ? hi[b] : hi[a], g_ = sw_ ? hi[a] : hi[b]; \
                         const uint32_t l_ = sw_ ? lo[b] : lo[a], k_ = sw_ ? lo[a] : lo[b]; hi[a] = h_; hi[b] = g_; lo[a] = l_; lo[b] = k_; } while (0)

// The reference's rescale of a context of at most eight states, in one lane, for OrderFall == 0 (Model.cpp:246-280):
// found state to the front, frequencies halved, stable insertion sort by the halved frequencies, SummFreq rebuilt, the
// found state's bonus.  Returns false -- and changes nothing -- when a state would drop out (frequency 1 -> 0): that path
// frees units (ShrinkUnits / FreeUnits) and stays with the serial code.
FS_DEV bool lane_rescale(uint32_t (&sf)[8], uint32_t (&sc)[8], uint32_t ns, uint32_t kf, uint32_t& summ, uint32_t& flags)
{
    uint32_t hi[8], lo[8];
    uint32_t sumOld = 0, sumNew = 0, f0 = 0; bool zeros = false, hiAny = false;
    #pragma unroll
    for (uint32_t j = 0; j < 8u; ++j) {
        const uint32_t f = sf[j] >> 8, sy = sf[j] & 0xFFu, nf = f >> 1;
        const bool valid = j <= ns, isF = j == kf;
        const uint32_t r = isF ? 0u : (j < kf ? j + 1u : j);                 // place after the move-to-front
        if (valid) { sumOld += f; sumNew += nf; }
        if (valid && isF) f0 = f;
        if (valid && !isF && nf == 0u) zeros = true;
        if (valid && !isF && nf != 0u && sy >= 0x40u) hiAny = true;
        const uint32_t key = !valid ? 0u : (isF ? 0xFFFFu : ((nf << 4) | (15u - r)));
        hi[j] = (key << 16) | (nf << 8) | sy; lo[j] = sc[j];
    }
    if (zeros) return false;
    // 19 compare-exchanges sort eight keys (keys are distinct: the place `r` is part of them)
    FS_CE(0, 1); FS_CE(2, 3); FS_CE(4, 5); FS_CE(6, 7);
    FS_CE(0, 2); FS_CE(1, 3); FS_CE(4, 6); FS_CE(5, 7);
    FS_CE(1, 2); FS_CE(5, 6); FS_CE(0, 4); FS_CE(3, 7);
    FS_CE(1, 5); FS_CE(2, 6);
    FS_CE(1, 4); FS_CE(3, 6);
    FS_CE(2, 4); FS_CE(3, 5);
    FS_CE(3, 4);
    #pragma unroll
    for (uint32_t j = 0; j < 8u; ++j) { sf[j] = hi[j] & 0xFFFFu; sc[j] = lo[j]; }
    const uint32_t escFreq = summ - sumOld, nf0 = f0 >> 1;
    uint32_t s = sumNew + ((escFreq + 1u) >> 1), a;
    if ((flags & 0x04u) == 0u) {
        const uint32_t sfm = summ - escFreq;
        a = sfm - f0;
        a |= (uint32_t)(a == 0u);                                        // (lanes that only ride along may hold anything)
        a = (f0 * s - sfm * nf0 + a - 1u) / a;
        a = a < 2u ? 2u : (a > (uint32_t)MAX_FREQ / 2u - 18u ? (uint32_t)MAX_FREQ / 2u - 18u : a);
    } else a = 2u;
    sf[0] = (sf[0] & 0xFFu) | ((nf0 + a) << 8);
    summ = s + a;
    flags = (flags & 0x14u) | (hiAny ? 0x08u : 0u) | 0x04u;
    return true;
}

// One window at position `pos` (the serial state is at the top of its loop with OrderFall == 0 and MinContext ==
// MaxContext).  Returns the number of symbols coded, 0 if the first position is not a plain hit.  On return > 0 the model
// memory, the coder, PrevSuccess, MaxContext and `hist` are exactly what the serial walk would have left.
FS_DEV uint32_t window_step(Coder& m, fs_cgptr in, uint32_t n, uint32_t pos, uint32_t MinContext, uint32_t& hist)
{
    const uint32_t lane = (uint32_t)FS_LANE();
    const uint32_t W = n - pos < (uint32_t)FS_WAVE ? n - pos : (uint32_t)FS_WAVE;
    const uint32_t q = pos + (lane < W ? lane : W - 1u);
    // the four bytes in front of the position and the position's own byte, from two aligned words
    const uint32_t a0 = (q - 4u) & ~3u, sh8 = 8u * ((q - 4u) & 3u);
    const uint32_t d0 = *(fs_cgptr32)(in + a0), d1 = *(fs_cgptr32)(in + a0 + 4u);
    const uint32_t key = sh8 ? ((d0 >> sh8) | (d1 << (32u - sh8))) : d0;
    const uint32_t sym = (d1 >> sh8) & 0xFFu;
    uint32_t addr;
    {
        fs_cgptr32 e = (fs_cgptr32)(m.hb + 1u + HINT_OFF + 8u * hint_slot(key));
        const uint32_t k0 = e[0], c0 = e[1];
        addr = k0 == key ? c0 : 0u;
    }
    if (lane == 0u) addr = MinContext;
    const uint32_t unitsStart = m.UnitsStart;
    bool ok = lane < W && addr >= unitsStart && addr <= SA_SIZE - 11u && ((addr - 1u) & 3u) == 0u;
    const uint32_t la = ok ? addr : MinContext;                   // lanes without a usable hint fetch somewhere harmless
    uint32_t r0, r1;
    { fs_cgptr32 p = (fs_cgptr32)HP(la); r0 = p[0]; r1 = p[1]; }
    const uint32_t ns = r0 & 0xFFu, stats = r1;
    ok = ok && ns >= 1u && ns <= WIN_MAX_NS && stats >= unitsStart && stats <= SA_SIZE - 47u && ((stats - 1u) & 3u) == 0u;
    const uint32_t ls = ok ? stats : la;
    FS_STAT_ADD(m.sh->winStats[0], 1u);

    uint32_t L = W;
    for (;;) {
        uint32_t sf[8], sc[8];
        {   // eight states = 48 bytes = twelve words; state j lives at byte 6 j
            fs_cgptr32 p = (fs_cgptr32)HP(ls);
            uint32_t w[12];
            #pragma unroll
            for (int i = 0; i < 12; ++i) w[i] = p[i];
            #pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t x = w[3 * t], y = w[3 * t + 1], z = w[3 * t + 2];
                sf[2 * t] = x & 0xFFFFu; sc[2 * t] = (x >> 16) | (y << 16);
                sf[2 * t + 1] = y >> 16; sc[2 * t + 1] = z;
            }
        }
        uint32_t summ = r0 >> 16, flags = (r0 >> 8) & 0xFFu;
        uint32_t k = 8u;
        #pragma unroll
        for (int j = 7; j >= 0; --j) if ((uint32_t)j <= ns && (sf[j] & 0xFFu) == sym) k = (uint32_t)j;
        uint32_t succ = 0;
        #pragma unroll
        for (int j = 0; j < 8; ++j) if ((uint32_t)j == k) succ = sc[j];
        // plain hit, given that the context is the right one; the chain proves the contexts
        const bool plain = ok && k < 8u && succ >= unitsStart;
        const uint32_t prevSucc = fs_bperm(succ, (lane + 63u) & 63u);
        const bool link = lane == 0u || addr == prevSucc;
        const uint64_t good = fs_ballot(plain && link && lane < L);
        const uint32_t lead = ~good ? fs_ctz64(~good) : 64u;
        L = lead < L ? lead : L;
        if (L == 0u) return 0u;

        // positions that share a context: rank among them, the lane before, and whether this is the last one
        uint32_t rank = 0, prevLane = 0; bool last = true;
        for (uint32_t j = 0; j < L; ++j) {
            const uint32_t a = fs_readlane(addr, j);
            const bool same = a == addr;
            if (same && j < lane) { ++rank; prevLane = j; }
            if (same && j > lane) last = false;
        }
        uint32_t tA = 0, tM = 0, ps = 0;
        uint64_t cutMask = 0;
        uint32_t rounds = 0;
        for (uint32_t r = 0;; ++r) {
            const bool act = lane < L && rank == r;
            if (fs_ballot(act) == 0ull) break;
            ++rounds;
            if (r > 0u) {   // the state of the context as the previous position of the same context left it
                #pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t a = fs_bperm(sf[j], prevLane), b = fs_bperm(sc[j], prevLane);
                    if (act) { sf[j] = a; sc[j] = b; }
                }
                const uint32_t a = fs_bperm(summ | (flags << 16), prevLane);
                if (act) { summ = a & 0xFFFFu; flags = a >> 16; }
            }
            // encodeSymbol1 + update1 on the lane's copy (Model.cpp:447-481)
            uint32_t kk = 8u;
            #pragma unroll
            for (int j = 7; j >= 0; --j) if ((uint32_t)j <= ns && (sf[j] & 0xFFu) == sym) kk = (uint32_t)j;
            uint32_t loCnt = 0, f = 0, fPrev = 0;
            #pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t fj = sf[j] >> 8;
                if ((uint32_t)j < kk) loCnt += fj;
                if ((uint32_t)j == kk) f = fj;
                if ((uint32_t)j + 1u == kk) fPrev = fj;
            }
            bool lost = act && kk >= 8u;                              // cannot happen while nothing drops out; never trust it
            const uint32_t nf = f + 4u;
            const bool doSwap = kk != 0u && kk < 8u && nf > fPrev;
            const bool resc = act && !lost && nf > (uint32_t)MAX_FREQ && (kk == 0u || doSwap);
            if (act && !lost) {
                const Recip rc = recip_make(summ);
                tA = loCnt | (f << 16) | (rc.l << 24); tM = rc.mul;
                ps = (kk == 0u && 2u * f > summ) ? 1u : 0u;
                #pragma unroll
                for (int j = 0; j < 8; ++j) if ((uint32_t)j == kk) sf[j] = (sf[j] & 0xFFu) | (nf << 8);
                #pragma unroll
                for (int j = 1; j < 8; ++j)
                    if (doSwap && (uint32_t)j == kk) { const uint32_t a = sf[j], b = sc[j]; sf[j] = sf[j - 1]; sc[j] = sc[j - 1]; sf[j - 1] = a; sc[j - 1] = b; }
                summ += 4u;
            }
            bool cut = lost;
            if (fs_ballot(resc) != 0ull) {
                uint32_t sf2[8], sc2[8], summ2 = summ, flags2 = flags;
                #pragma unroll
                for (int j = 0; j < 8; ++j) { sf2[j] = sf[j]; sc2[j] = sc[j]; }
                const bool done = lane_rescale(sf2, sc2, ns, doSwap ? kk - 1u : kk, summ2, flags2);
                if (resc && done) {
                    #pragma unroll
                    for (int j = 0; j < 8; ++j) { sf[j] = sf2[j]; sc[j] = sc2[j]; }
                    summ = summ2; flags = flags2;
                }
                if (resc && !done) cut = true;                     // a state drops out: the serial path takes this symbol
            }
            cutMask |= fs_ballot(cut);
        }
        FS_STAT_ADD(m.sh->winStats[3], rounds);
        if (cutMask != 0ull) {                                        // nothing has been stored yet: shorten the window and redo it
            const uint32_t at = fs_ctz64(cutMask);
            FS_STAT_ADD(m.sh->winStats[4], 1u);
            L = at;                                                   // at < L: every redo is strictly shorter
            if (L == 0u) return 0u;
            continue;
        }

        // commit: the last position of each context writes the list and the record word back
        if (lane < L && last) {
            fs_gptr32 p = (fs_gptr32)HP(stats);
            const uint32_t nst = ns + 1u, full = (3u * nst) >> 1;
            uint32_t w[12];
            #pragma unroll
            for (int t = 0; t < 4; ++t) {
                w[3 * t] = sf[2 * t] | (sc[2 * t] << 16);
                w[3 * t + 1] = (sc[2 * t] >> 16) | (sf[2 * t + 1] << 16);
                w[3 * t + 2] = sc[2 * t + 1];
            }
            #pragma unroll
            for (int i = 0; i < 12; ++i) if ((uint32_t)i < full) p[i] = w[i];
            if (nst & 1u) {
                uint32_t tail = 0;
                #pragma unroll
                for (int i = 0; i < 12; ++i) if ((uint32_t)i == full) tail = w[i];
                *(fs_gptr16)(HP(stats) + 4u * full) = (uint16_t)tail;
            }
            *(fs_gptr32)HP(addr) = ns | (flags << 8) | (summ << 16);
        }
        FS_WAVE_SYNC();

        // the range coder, in stream order (Coder.hpp:13-17 + the normalisation of Model.cpp:580)
        for (uint32_t i = 0; i < L; ++i) {
            const uint32_t A = FS_UNI(fs_readlane(tA, i)), M = FS_UNI(fs_readlane(tM, i));
            const uint32_t rr = recip_div(m.range, M, A >> 24);
            m.low += (A & 0xFFFFu) * rr; m.range = rr * ((A >> 16) & 0xFFu);
            rc_normalize(m);
        }
        m.PrevSuccess = FS_UNI(fs_readlane(ps, L - 1u));
        m.MaxContext = FS_UNI(fs_readlane(succ, L - 1u));
        const uint32_t kl = FS_UNI(fs_readlane(key, L - 1u)), sl = FS_UNI(fs_readlane(sym, L - 1u));
        hist = (kl >> 8) | (sl << 24);
        FS_STAT_ADD(m.sh->winStats[1], 1u);
        FS_STAT_ADD(m.sh->winStats[2], L);
        return L;
    }
}
#undef FS_CE
