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

// ---- packed state list of one context in a lane: N states as bytes, no register arrays indexed at run time ----
// S = symbols, F = frequencies (state j in byte j of the word), P = nibble j names the original slot whose successor field
// belongs to the state now at place j (the 32-bit successors themselves never move: they are gathered through P when the
// list is written back).  N = 8: 64-bit words (contexts of up to eight states).  (Round 3 also had 32-bit forms for windows
// whose contexts all have at most four states: half the ALU work, and SLOWER -- 1 012 ms against 976 ms on a lone 7 M-symbol
// stream, profiles/r03_narrow_windows.txt: the window is bound by its chain of LDS and ballot round trips, and the extra
// wave-wide vote costs more than the halved arithmetic saves.  They left the tree in round 4.)
template <int N> struct PackedT;
template <> struct PackedT<8> { typedef uint64_t W; uint64_t S, F; uint32_t P; };
typedef PackedT<8> Packed;

FS_DEV uint32_t fs_sum_bytes(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sad_u8(x, 0u, 0u);                     // v_sad_u8: sum of the four bytes
#else
    return (x & 0xFFu) + ((x >> 8) & 0xFFu) + ((x >> 16) & 0xFFu) + (x >> 24);
#endif
}
FS_DEV uint32_t fs_sum_bytes_w(uint64_t x) { return fs_sum_bytes((uint32_t)x) + fs_sum_bytes((uint32_t)(x >> 32)); }
FS_DEV uint32_t fs_sum_bytes_w(uint32_t x) { return fs_sum_bytes(x); }
FS_DEV uint32_t fs_ctz_w(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }
FS_DEV uint32_t fs_ctz_w(uint32_t x) { return (uint32_t)__builtin_ctz(x); }
template <class W> FS_DEV W fs_rep8(uint32_t b) { return (W)(0x0101010101010101ull * b); }      // the byte in every byte of the word
// the bytes of the states 0 .. ns
template <int N> FS_DEV typename PackedT<N>::W fs_valid_bytes(uint32_t ns)
{ typedef typename PackedT<N>::W W; return ns >= (uint32_t)N - 1u ? (W)~(W)0 : (W)(((W)1 << (8u * (ns + 1u))) - (W)1); }
FS_DEV uint32_t fs_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
FS_DEV uint32_t fs_umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

// place of `sym` among the states 0..ns (8 = absent)
template <int N> FS_DEV uint32_t packed_find(const PackedT<N>& c, uint32_t ns, uint32_t sym)
{
    typedef typename PackedT<N>::W W;
    const W x = c.S ^ fs_rep8<W>(sym);
    // a byte of x is zero <=> that state has the symbol; the lowest flagged byte of the classic test is always exact
    W z = (W)(x - fs_rep8<W>(1u)) & (W)~x & fs_rep8<W>(0x80u);
    z &= fs_valid_bytes<N>(ns);
    return z ? fs_ctz_w(z) >> 3 : 8u;
}

// The reference's rescale for OrderFall == 0 (Model.cpp:246-280) on a packed list: found state (place kf) to the front,
// frequencies halved, stable insertion sort by the halved frequencies -- a 19-step min/max network on one word per
// state --, SummFreq rebuilt, the found state's bonus.  States whose frequency was 1 drop out (nsOut = the NumStats field behind
// the rescale; the caller shrinks the units when the window is committed: win_write_back).  Returns false and changes nothing
// when only the found state would be left: that context turns binary (FreeUnits), which stays with the serial code.
// K = 4: every context that rescales has at most four states (the caller checks, wave-wide): a five-step network and half
// the packing
template <int K, int N> FS_DEV bool packed_rescale(PackedT<N>& c, uint32_t ns, uint32_t kf, uint32_t& summ, uint32_t& flags, uint32_t& nsOut)
{
    typedef typename PackedT<N>::W W;
    uint32_t key[K];
    uint32_t sumOld = 0, sumNew = 0, f0 = 0, zeros = 0; bool hiAny = false;
    #pragma unroll
    for (uint32_t j = 0; j < (uint32_t)K; ++j) {
        const uint32_t f = (uint32_t)(c.F >> (8u * j)) & 0xFFu, sy = (uint32_t)(c.S >> (8u * j)) & 0xFFu, pj = (c.P >> (4u * j)) & 0xFu, nf = f >> 1;
        const bool valid = j <= ns, isF = j == kf;
        const uint32_t r = isF ? 0u : (j < kf ? j + 1u : j);                 // place after the move-to-front
        if (valid) { sumOld += f; sumNew += nf; }
        if (valid && isF) f0 = f;
        if (valid && !isF && nf == 0u) ++zeros;
        if (valid && !isF && nf != 0u && sy >= 0x40u) hiAny = true;
        // descending order of the keys = the reference's order: halved frequency, then the earlier place; found state on top.  A state
        // that drops out (frequency 1 -> 0; Model.cpp:264-275) keeps a key below every state that stays and above the unused places
        key[j] = !valid ? 0u : (((isF ? 0xFFu : nf) << 24) | ((15u - r) << 20) | (nf << 12) | (sy << 4) | pj);
    }
    nsOut = ns - zeros;
    if (nsOut == 0u && zeros != 0u) return false;                           // one state left: the context turns binary (units freed) -- the serial path's business
    #define FS_CE(a, b) do { const uint32_t hi_ = fs_umax(key[a], key[b]), lo_ = fs_umin(key[a], key[b]); key[a] = hi_; key[b] = lo_; } while (0)
    if (K == 8) {
        FS_CE(0, 1); FS_CE(2, 3); FS_CE(4 % K, 5 % K); FS_CE(6 % K, 7 % K);
        FS_CE(0, 2); FS_CE(1, 3); FS_CE(4 % K, 6 % K); FS_CE(5 % K, 7 % K);
        FS_CE(1, 2); FS_CE(5 % K, 6 % K); FS_CE(0, 4 % K); FS_CE(3, 7 % K);
        FS_CE(1, 5 % K); FS_CE(2, 6 % K);
        FS_CE(1, 4 % K); FS_CE(3, 6 % K);
        FS_CE(2, 4 % K); FS_CE(3, 5 % K);
        FS_CE(3, 4 % K);
    } else {
        FS_CE(0, 1); FS_CE(2, 3); FS_CE(0, 2); FS_CE(1, 3); FS_CE(1, 2);
    }
    #undef FS_CE
    const uint32_t escFreq = summ - sumOld + zeros, nf0 = f0 >> 1;          // (EscFreq += the states that dropped out: Model.cpp:268)
    uint32_t s = sumNew + ((escFreq + 1u) >> 1), a;
    if ((flags & 0x04u) == 0u) {
        const uint32_t sfm = summ - escFreq;
        a = sfm - f0;
        a |= (uint32_t)(a == 0u);                                        // (lanes that only ride along may hold anything)
        a = (f0 * s - sfm * nf0 + a - 1u) / a;
        a = a < 2u ? 2u : (a > (uint32_t)MAX_FREQ / 2u - 18u ? (uint32_t)MAX_FREQ / 2u - 18u : a);
    } else a = 2u;
    W S = 0, F = 0; uint32_t P = 0;
    #pragma unroll
    for (uint32_t j = 0; j < (uint32_t)K; ++j) {
        const bool stays = j <= nsOut;                                    // the places behind the last state that stays hold nothing
        S |= (W)(stays ? (key[j] >> 4) & 0xFFu : 0u) << (8u * j);
        F |= (W)(!stays ? 0u : (j == 0u ? nf0 + a : ((key[j] >> 12) & 0xFFu))) << (8u * j);
        P |= (key[j] & 0xFu) << (4u * j);
    }
    c.S = S; c.F = F; c.P = P;
    summ = s + a;
    flags = (flags & 0x14u) | (hiAny ? 0x08u : 0u) | 0x04u;
    return true;
}

// The usual rescale needs no sorting at all: the state that reaches MAX_FREQ is the one at the front (the found state stays
// there), and halving keeps the order of the others unless two of them were out of order before (a single +4 may lift a
// state over TWO predecessors, and only one swap is made).  ok(): found state at place 0, halved frequencies of the places
// 1 .. ns not increasing -- then the network would leave every state where it is, and the rest is byte arithmetic on the
// packed words.  Same results as packed_rescale on that domain (checked against it on the lock-step emulation).
template <int N> FS_DEV bool packed_rescale_quick_ok(const PackedT<N>& c, uint32_t ns, uint32_t kf)
{
    typedef typename PackedT<N>::W W;
    const W vm = fs_valid_bytes<N>(ns);                                                    // the bytes of the valid states
    const W nf = (W)(c.F >> 1) & fs_rep8<W>(0x7Fu) & vm;
    const W ge = (W)((nf | fs_rep8<W>(0x80u)) - (W)(nf >> 8)) & fs_rep8<W>(0x80u);          // byte j: nf[j] >= nf[j+1] (both below 128: no borrow crosses a byte)
    const W need = (W)(vm >> 8) & (W)~(W)0xFFu & fs_rep8<W>(0x80u);                         // places 1 .. ns-1
    // ... and no state but the found one drops to zero (those lists are compacted by the network form)
    const W zx = nf | (W)0xFFu | (W)~vm;
    const bool zeros = ((W)(zx - fs_rep8<W>(1u)) & (W)~zx & fs_rep8<W>(0x80u)) != (W)0;
    return kf == 0u && !zeros && (ge & need) == need;
}

template <int N> FS_DEV bool packed_rescale_quick(PackedT<N>& c, uint32_t ns, uint32_t& summ, uint32_t& flags, bool live)
{
    typedef typename PackedT<N>::W W;
    const W vm = fs_valid_bytes<N>(ns);
    const W nf = (W)(c.F >> 1) & fs_rep8<W>(0x7Fu) & vm;
    // a state other than the found one that would drop to zero: the serial path's business (no early return: every lane
    // takes part in the vote below)
    const W zx = nf | (W)0xFFu | (W)~vm;
    const bool zeros = ((W)(zx - fs_rep8<W>(1u)) & (W)~zx & fs_rep8<W>(0x80u)) != (W)0;
    const W fv = c.F & vm;
    const uint32_t sumOld = fs_sum_bytes_w(fv), sumNew = fs_sum_bytes_w(nf);
    const uint32_t f0 = (uint32_t)c.F & 0xFFu, nf0 = f0 >> 1;
    const bool hiAny = (c.S & vm & (W)~(W)0xFFu & fs_rep8<W>(0xC0u)) != (W)0;
    const uint32_t escFreq = summ - sumOld;
    uint32_t s = sumNew + ((escFreq + 1u) >> 1), a = 2u;
    // (a context's FIRST rescale sets the found state's bonus by a division -- forty instructions the wave only walks
    // through when a lane that really rescales needs them; lanes that ride along keep whatever they compute)
    if (fs_ballot(live && !zeros && (flags & 0x04u) == 0u) != 0ull) {
        const uint32_t sfm = summ - escFreq;
        uint32_t d = sfm - f0;
        d |= (uint32_t)(d == 0u);
        d = (f0 * s - sfm * nf0 + d - 1u) / d;
        d = d < 2u ? 2u : (d > (uint32_t)MAX_FREQ / 2u - 18u ? (uint32_t)MAX_FREQ / 2u - 18u : d);
        if ((flags & 0x04u) == 0u) a = d;
    }
    if (zeros) return false;
    c.F = (W)(nf & (W)~(W)0xFFu) | (W)(nf0 + a);
    c.S &= vm;
    c.P &= ns >= (uint32_t)N - 1u ? (N == 8 ? ~0u : 0xFFFFu) : ((1u << (4u * (ns + 1u))) - 1u);
    summ = s + a;
    flags = (flags & 0x14u) | (hiAny ? 0x08u : 0u) | 0x04u;
    return true;
}

// ---- a window in three steps: fetch (per position, independent of where the window starts), solve (which positions form
// the window, who owns which context, every position's price), commit (lists and records back, the prices to the coder).
// One wave does all three in a row (window_step below).
//
// Window = the lanes [s, E): lane i holds position base + i.  Positions that share a context are taken by the lane of the
// FIRST of them (the owner): it keeps the context's states in its registers and walks its positions in stream order, one
// per round, so the rounds of a window are as many as the most popular context has positions.
#define FS_PROF_ACC_W(w, t0) FS_PROF_ACC(w, t0)
struct WinHead {                   // per lane: the position, its context's record
    uint32_t key, sym, addr, r0, stats, ns, la; bool ok;
    uint32_t W;                     // positions left in the stream from `base` on, at most 64
};
template <int N> struct WinFetch : WinHead {   // ... and its state list, as fetched
    PackedT<N> c; uint32_t sc[N];   // symbols, frequencies, successors
    uint32_t k, succ; bool plain;   // place of the position's symbol in the list, its successor; a plain hit if the context is the right one
    bool link;                      // this lane's context is the successor the lane before found (lane 0: by definition)
};
template <int N> struct WinSolved {            // per lane, after the rounds
    uint32_t ownerLane; bool owner;
    PackedT<N> c; uint32_t summ, flags;        // an owner's context after all its positions
    uint32_t ns; bool dropped; uint32_t dropPos;   // ... its NumStats field then; a rescale of the window let states drop out (one per context and window at most), at this position
    uint32_t tA, tM;                // the position's price: cumulative frequency | frequency << 16 | PrevSuccess << 23 ; the total
};

// lane0ctx != 0: the context of position `base` is known (the serial walk is there); otherwise every lane asks the hint table
FS_DEV void win_fetch_head(Coder& m, fs_cgptr in, uint32_t n, uint32_t base, uint32_t lane0ctx, WinHead& f)
{
    const uint32_t lane = (uint32_t)FS_LANE();
    f.W = n - base < (uint32_t)FS_WAVE ? n - base : (uint32_t)FS_WAVE;
    const uint32_t q = base + (lane < f.W ? lane : f.W - 1u);
    // the four bytes in front of the position and the position's own byte, from two aligned words
    const uint32_t a0 = (q - 4u) & ~3u, sh8 = 8u * ((q - 4u) & 3u);
    const uint32_t d0 = *(fs_cgptr32)(in + a0), d1 = *(fs_cgptr32)(in + a0 + 4u);
    f.key = sh8 ? ((d0 >> sh8) | (d1 << (32u - sh8))) : d0;
    f.sym = (d1 >> sh8) & 0xFFu;
    uint32_t addr;
    {
        fs_cgptr32 e = (fs_cgptr32)(m.hb + 1u + HINT_OFF + 8u * hint_slot(f.key));
        const uint32_t k0 = e[0], c0 = e[1];
        addr = k0 == f.key ? c0 : 0u;
    }
    if (lane == 0u && lane0ctx != 0u) addr = lane0ctx;
    f.addr = addr;
    const uint32_t unitsStart = m.UnitsStart;
    bool ok = lane < f.W && addr >= unitsStart && addr <= SA_SIZE - 11u && ((addr - 1u) & 3u) == 0u;
    const uint32_t safe = lane0ctx != 0u ? lane0ctx : unitsStart;     // lanes without a usable hint fetch somewhere harmless
    const uint32_t la = ok ? addr : safe;
    uint32_t r0, r1;
    { fs_cgptr32 p = (fs_cgptr32)HP(la); r0 = p[0]; r1 = p[1]; }
    const uint32_t ns = r0 & 0xFFu, stats = r1;
    ok = ok && ns >= 1u && ns <= WIN_MAX_NS && stats >= unitsStart && stats <= SA_SIZE - 47u && ((stats - 1u) & 3u) == 0u;
    f.r0 = r0; f.ns = ns; f.stats = stats; f.ok = ok; f.la = la;
}
template <int N> FS_DEV void win_fetch_list(Coder& m, const WinHead& h, WinFetch<N>& f)
{
    typedef typename PackedT<N>::W W;
    const uint32_t lane = (uint32_t)FS_LANE();
    static_cast<WinHead&>(f) = h;
    const uint32_t ls = h.ok ? h.stats : h.la;
    {   // N states = 6 N bytes = 3 N / 2 words; state j lives at byte 6 j
        fs_cgptr32 p = (fs_cgptr32)HP(ls);
        uint32_t w[3 * N / 2];
        #pragma unroll
        for (int i = 0; i < 3 * N / 2; ++i) w[i] = p[i];
        f.c.S = 0; f.c.F = 0; f.c.P = N == 8 ? 0x76543210u : 0x3210u;
        #pragma unroll
        for (int t = 0; t < N / 2; ++t) {
            const uint32_t x = w[3 * t], y = w[3 * t + 1], z = w[3 * t + 2];
            f.c.S |= (W)((x & 0xFFu) | ((y >> 8) & 0xFF00u)) << (16 * t);
            f.c.F |= (W)(((x >> 8) & 0xFFu) | ((y >> 16) & 0xFF00u)) << (16 * t);
            f.sc[2 * t] = (x >> 16) | (y << 16); f.sc[2 * t + 1] = z;
        }
    }
    f.k = packed_find<N>(f.c, h.ns, h.sym);
    uint32_t succ = 0;
    #pragma unroll
    for (int j = 0; j < N; ++j) if ((uint32_t)j == f.k) succ = f.sc[j];
    f.succ = succ;
    // plain hit, given that the context is the right one; the chain proves the contexts
    f.plain = h.ok && f.k < 8u && succ >= m.UnitsStart;
    const uint32_t prevSucc = fs_bperm(succ, (lane + 63u) & 63u);
    f.link = lane == 0u || h.addr == prevSucc;
}
// The window [s, E) of a fetch: E = the first lane from s on that is not a plain hit linked to the lane before it (lane s
// itself needs no link: its context is known, or is checked by the caller), further shortened where a round finds a
// position for the serial path (a rescale that drops a state).  Returns E; E <= s: no window.
template <int N> FS_DEV uint32_t win_solve(Coder& m, fs_cgptr in, uint32_t n, uint32_t base, const WinFetch<N>& f, uint32_t s, WinSolved<N>& o, uint64_t& tp)
{
    typedef typename PackedT<N>::W W;
    const uint32_t lane = (uint32_t)FS_LANE();
    const uint32_t ns = f.ns, sym = f.sym, addr = f.addr, k = f.k;
    uint32_t E;
    {
        const uint64_t from = s >= 64u ? 0ull : ~0ull << s;
        const uint64_t good = fs_ballot(f.plain && (f.link || lane == s) && lane < f.W);
        const uint64_t bad = ~good & from;
        E = bad ? fs_ctz64(bad) : 64u;
        if (s >= 64u) E = s;
    }
    o.ownerLane = lane; o.owner = false; o.c = f.c; o.summ = f.r0 >> 16; o.flags = (f.r0 >> 8) & 0xFFu; o.tA = 0; o.tM = 0; o.ns = f.ns; o.dropped = false; o.dropPos = 0;
    FS_PROF_ACC_W(m.sh->winStats[9], tp);                            // state lists, find, chain
    if (E <= s) return E;

    // The owner of a context = the first window position that has it.  A 512-slot table in LDS names, per hash
    // slot, the lowest lane that wrote to it; a lane whose slot was won by a lane with ANOTHER context (a collision)
    // is resolved by the loop below, one step per distinct context among the colliding lanes -- rare.  Then every
    // position sets its bit in its owner's mask.
    bool inWin = lane >= s && lane < E;
    uint32_t ownerLane = lane;
    {
        #pragma unroll
        for (uint32_t i = 0; i < 8u; ++i) m.sh->winTab[64u * i + lane] = 0u;
        m.sh->winMask[2u * lane] = 0u; m.sh->winMask[2u * lane + 1u] = 0u;
        if (lane == 0u) m.sh->winCut = 64u;
        FS_WAVE_SYNC();
        const uint32_t h = (addr * 0x9E3779B1u) >> 23;
        if (inWin) FS_LDS_MAX(m.sh->winTab[h], 64u - lane);
        FS_WAVE_SYNC();
        const uint32_t w = (64u - m.sh->winTab[h]) & 63u;
        const uint32_t aw = fs_bperm(addr, w);
        if (inWin && aw == addr) ownerLane = w;
        for (uint64_t todo = fs_ballot(inWin && aw != addr); todo != 0ull;) {
            const uint32_t j = fs_ctz64(todo), a = fs_readlane(addr, j);
            const bool mine = inWin && aw != addr && addr == a;
            if (mine) ownerLane = j;
            todo &= ~fs_ballot(mine);
        }
        if (inWin) FS_LDS_OR(m.sh->winMask[2u * ownerLane + (lane >> 5)], 1u << (lane & 31u));
        FS_WAVE_SYNC();
    }
    bool owner = inWin && ownerLane == lane;
    uint32_t rlo = 0u, rhi = 0u;                                   // an owner's positions still to do
    FS_PROF_ACC_W(m.sh->winStats[10], tp);                           // context sets
    // Input read-ahead.  The stream is read once, so every new cache line of it is a trip to HBM (~900 clocks against
    // ~200 for an L2 hit) -- paid by the first loads of a window and by the serial path's byte fetches.  A kilobyte
    // ahead of the window, sixteen lines at a time, is requested HERE: the rounds that follow use no global memory,
    // so the trip is over before the next load of this wave has to wait for it (loads complete in issue order).
    uint32_t ahead = 0;
    const bool pull = FS_UNI((uint32_t)(base + 1024u >= m.inAhead));
    if (pull) {
        const uint32_t want = base + 1024u + 16u * lane, last = (n - 4u) & ~3u;
        ahead = *(fs_cgptr32)(in + (want < last ? want & ~3u : last));
        m.inAhead = base + 2048u;
    }
    uint32_t rounds = 0;
    // Closed form.  While a context neither swaps two states nor crosses MAX_FREQ, its states stay where they were
    // fetched and every hit just adds 4 to one frequency and to the total -- so a position can price ITSELF from the
    // list it fetched and three counts over the earlier positions of its context (all of them / those on its own
    // state / those on states in front of it), taken from bit masks: the context's positions (its owner's mask) and
    // the bit-sliced places of all 64 positions.  The first position of a context whose step would swap or rescale,
    // and everything of that context behind it, is left to the rounds below; the owner accounts for what went
    // before (frequencies and total) and keeps the rest of its mask.
    uint32_t cfA = 0, cfM = 0; bool cfDone = false;
    uint64_t cfSm = 0, cfKb0 = 0, cfKb1 = 0, cfKb2 = 0; uint32_t cfFirstBad = 64u;
    const uint32_t summ0 = f.r0 >> 16, flags0 = (f.r0 >> 8) & 0xFFu;
    {
        const uint32_t smLo = m.sh->winMask[2u * ownerLane], smHi = m.sh->winMask[2u * ownerLane + 1u];
        const uint64_t sm = inWin ? ((uint64_t)smHi << 32) | smLo : 0ull;
        const uint64_t kb0 = fs_ballot(inWin && (k & 1u) != 0u), kb1 = fs_ballot(inWin && (k & 2u) != 0u), kb2 = N == 8 ? fs_ballot(inWin && (k & 4u) != 0u) : 0ull;      // (N = 4: places 0..3)
        const uint64_t earlier = sm & ((1ull << lane) - 1ull);
        const uint32_t kq = k & 7u, kp = (kq - 1u) & 7u;
        const uint64_t e0 = (kq & 1u) ? kb0 : ~kb0, e1 = (kq & 2u) ? kb1 : ~kb1, e2 = (kq & 4u) ? kb2 : ~kb2;
        const uint64_t eqK = e0 & e1 & e2;
        const uint64_t eqP = ((kp & 1u) ? kb0 : ~kb0) & ((kp & 2u) ? kb1 : ~kb1) & ((kp & 4u) ? kb2 : ~kb2);
        const uint64_t ltK = ((kq & 4u) ? ~kb2 : 0ull) | (e2 & (((kq & 2u) ? ~kb1 : 0ull) | (e1 & ((kq & 1u) ? ~kb0 : 0ull))));
        const uint32_t cAll = fs_popc64(earlier), cSame = fs_popc64(earlier & eqK), cPrev = fs_popc64(earlier & eqP), cBelow = fs_popc64(earlier & ltK);
        const uint32_t k8 = 8u * kq;
        const uint32_t fr = ((uint32_t)(f.c.F >> k8) & 0xFFu) + 4u * cSame;
        const uint32_t fPrev = kq ? ((uint32_t)(f.c.F >> ((k8 - 8u) & (8u * N - 1u))) & 0xFFu) + 4u * cPrev : 0xFFFFu;
        const bool bad = inWin && (fr + 4u > fPrev || fr + 4u > (uint32_t)MAX_FREQ);
        const uint64_t badSet = fs_ballot(bad) & sm;               // of my context
        const uint32_t firstBad = badSet ? fs_ctz64(badSet) : 64u;
        cfDone = inWin && lane < firstBad;
        const W below = f.c.F & (W)(((W)1 << k8) - (W)1);
        const uint32_t tot = summ0 + 4u * cAll;
        cfA = (fs_sum_bytes_w(below) + 4u * cBelow) | (fr << 16) | ((kq == 0u && 2u * fr > tot) ? (1u << 23) : 0u);
        cfM = tot;
        cfSm = sm; cfFirstBad = firstBad; cfKb0 = kb0; cfKb1 = kb1; cfKb2 = kb2;
    }
    if (fs_ballot(inWin && !cfDone) == 0ull) FS_STAT_ADD(m.sh->winStats[5], 1u);     // a window without a single round
    // The owners' part, and the rounds.  When a round finds that a position must go to the serial path (a rescale that
    // drops a state), the window ends in front of it -- and only THIS part is done again for the shorter window: the
    // lists, the chain, the context sets and the closed-form prices of the positions that stay do not depend on
    // the positions that go.
    PackedT<N> c; uint32_t summ, flags;
    uint32_t nsCur, dropPos; bool dropped;       // an owner's NumStats field as the rounds go (a rescale may let states drop out: once per context and window)
    for (;;) {
        inWin = lane >= s && lane < E; owner = inWin && ownerLane == lane;
        c = f.c; summ = summ0; flags = flags0; rlo = rhi = 0u; rounds = 0;
        nsCur = ns; dropped = false; dropPos = 0u;
        {   // the owner: what its context's finished positions added, and what is left for the rounds
            const uint64_t lim = (E >= 64u ? ~0ull : (1ull << E) - 1ull) & (~0ull << s);
            const uint64_t doneSet = (cfFirstBad < 64u ? cfSm & ((1ull << cfFirstBad) - 1ull) : cfSm) & lim;
            if (owner) {
                W add = 0;
                #pragma unroll
                for (uint32_t j = 0; j < (uint32_t)N; ++j) {
                    const uint64_t ej = ((j & 1u) ? cfKb0 : ~cfKb0) & ((j & 2u) ? cfKb1 : ~cfKb1) & ((j & 4u) ? cfKb2 : ~cfKb2);
                    add |= (W)(4u * fs_popc64(doneSet & ej)) << (8u * j);      // no byte overflows: every frequency stays <= MAX_FREQ
                }
                c.F += add; summ += 4u * fs_popc64(doneSet);
                const uint64_t rest = cfSm & lim & ~doneSet;
                rlo = (uint32_t)rest; rhi = (uint32_t)(rest >> 32);
            }
            if (lane == 0u) m.sh->winCut = 64u;
            FS_WAVE_SYNC();
        }
        // Full rounds: the contexts that swap or rescale (again without branches inside a round, except for the rare
        // rescale).  A lane that must hand a position to the serial path remembers it; the minimum is taken after the loop.
        uint32_t myCut = 64u;
        for (;;) {
            const bool act = (rlo | rhi) != 0u;
            if (fs_ballot(act) == 0ull) break;
            ++rounds;
            const uint32_t pLo = (uint32_t)__builtin_ctz(rlo | 0x80000000u), pHi = 32u + (uint32_t)__builtin_ctz(rhi | 0x80000000u);
            const uint32_t p = !act ? lane : (rlo ? pLo : pHi);
            { const uint32_t nlo = rlo & (rlo - 1u), nhi = rhi & (rhi - 1u); rhi = rlo == 0u ? nhi : rhi; rlo = nlo; }
            const uint32_t sy = fs_bperm(sym, p);
            // encodeSymbol1 + update1 on the owner's copy (Model.cpp:447-481)
            const uint32_t kk = packed_find<N>(c, nsCur, sy);
            const bool lost = act && kk >= 8u;                        // the symbol's state dropped out in a rescale of this window: an escape, the serial path's
            const bool go = act && !lost;
            const uint32_t k8 = 8u * (kk & 7u);
            const uint32_t fr = (uint32_t)(c.F >> k8) & 0xFFu, fPrev = (kk & 7u) ? (uint32_t)(c.F >> ((k8 - 8u) & (8u * N - 1u))) & 0xFFu : 0u;
            const W below = c.F & (W)(((W)1 << k8) - (W)1);
            const uint32_t loCnt = fs_sum_bytes_w(below);
            const uint32_t nf = fr + 4u;
            const bool doSwap = go && kk != 0u && nf > fPrev;
            const bool resc = go && nf > (uint32_t)MAX_FREQ && (kk == 0u || doSwap);
            // slot of position p: cumulative frequency | frequency << 16 | PrevSuccess << 23 ; the total
            const uint32_t at = go ? p : 64u + lane;
            m.sh->winA[at] = loCnt | (fr << 16) | ((kk == 0u && 2u * fr > summ) ? (1u << 23) : 0u);
            m.sh->winM[at] = summ;
            c.F += go ? (W)4 << k8 : (W)0;
            summ += go ? 4u : 0u;
            {   // states kk and kk-1 change places: symbol, frequency, successor tag (the differences are zero without a swap)
                const uint32_t j8 = (k8 - 8u) & (8u * N - 1u), j4 = (4u * kk - 4u) & 31u;
                const W dS = doSwap ? (W)((c.S >> j8) ^ (c.S >> k8)) & (W)0xFFu : (W)0, dF = doSwap ? (W)((c.F >> j8) ^ (c.F >> k8)) & (W)0xFFu : (W)0;
                c.S ^= (dS << j8) | (dS << k8); c.F ^= (dF << j8) | (dF << k8);
                const uint32_t dP = doSwap ? ((c.P >> j4) ^ (c.P >> ((j4 + 4u) & 31u))) & 0xFu : 0u;
                c.P ^= (dP << j4) | (dP << ((j4 + 4u) & 31u));
            }
#if defined(FS_SER_PROFILE)
            FS_STAT_ADD(m.sh->serStats[5], fs_popc64(fs_ballot(resc)));
            FS_STAT_ADD(m.sh->serStats[6], fs_popc64(fs_ballot(doSwap)));
            FS_STAT_ADD(m.sh->serStats[7], fs_popc64(fs_ballot(act)));
#endif
            bool cut = lost;
            if (fs_ballot(resc) != 0ull) {
                PackedT<N> c2 = c; uint32_t summ2 = summ, flags2 = flags;
                const uint32_t kf = doSwap ? kk - 1u : kk;
                bool done; uint32_t nsOut = nsCur;
                if (fs_ballot(resc && !packed_rescale_quick_ok<N>(c, nsCur, kf)) == 0ull) {
                    done = packed_rescale_quick<N>(c2, nsCur, summ2, flags2, resc);
#if defined(FS_SIMT_EMU)
                    {   // the lock-step emulation holds the short form against the network, lane by lane
                        PackedT<N> c3 = c; uint32_t summ3 = summ, flags3 = flags, ns3 = nsCur;
                        const bool done3 = packed_rescale<N, N>(c3, nsCur, kf, summ3, flags3, ns3);
                        if (resc && (done3 != done || ns3 != nsCur || (done && (c3.S != c2.S || c3.F != c2.F || c3.P != c2.P || summ3 != summ2 || flags3 != flags2)))) {
                            fprintf(stderr, "packed_rescale_quick differs from the network (ns %u)\n", nsCur); abort();
                        }
                        if (resc) simt_count_quick_rescale();
                    }
#endif
                } else done = fs_ballot(resc && nsCur > 3u) == 0ull ? packed_rescale<4, N>(c2, nsCur, kf, summ2, flags2, nsOut) : packed_rescale<(N == 8 ? 8 : 4), N>(c2, nsCur, kf, summ2, flags2, nsOut);
                // States dropped out (frequency 1 -> 0: Model.cpp:264-275).  The list is compact in the registers already; its units are
                // shrunk when the window is committed, in stream order with the other contexts' (win_write_back).  A second such rescale
                // of one context inside one window is left to the serial path: the reference would shrink its units twice, with other
                // contexts' blocks moving through the free lists in between.
                const bool drops = resc && done && nsOut != nsCur;
                if (resc && done && !(drops && dropped)) { c = c2; summ = summ2; flags = flags2; nsCur = nsOut; }
                if (drops && dropped) cut = true;
#if !defined(FS_WIN_PROFILE)
                { const uint32_t nDrops = fs_popc64(fs_ballot(drops && !dropped)); FS_STAT_ADD(m.sh->winStats[7], nDrops); }      // rescales that let states drop out inside a window (profile builds keep a clock in this word)
#endif
                if (drops && !dropped) { dropped = true; dropPos = p; }
                if (resc && !done) cut = true;                     // the context turns binary: the serial path takes this symbol
            }
            myCut = cut && p < myCut ? p : myCut;
            rlo = cut ? 0u : rlo; rhi = cut ? 0u : rhi;
        }
        if (fs_ballot(myCut < 64u) != 0ull) FS_LDS_MIN(m.sh->winCut, myCut);
        FS_KEEP(ahead);
        FS_WAVE_SYNC();
        FS_STAT_ADD(m.sh->winStats[3], rounds);
        FS_PROF_ACC_W(m.sh->winStats[11], tp);                           // rounds
        const uint32_t cutAt = FS_LDS_RD(m.sh->winCut);
        if (cutAt < E) {                                              // nothing has been stored yet: shorten the window and do the owners' part again
            FS_STAT_ADD(m.sh->winStats[4], 1u);
            E = cutAt;                                                // every redo is strictly shorter
            if (E <= s) return E;
            continue;
        }
        break;
    }
    inWin = lane >= s && lane < E;
    // every position's price: its own (closed form), or what its owner left in its slot
    o.tA = inWin ? (cfDone ? cfA : m.sh->winA[lane]) : 0u; o.tM = inWin ? (cfDone ? cfM : m.sh->winM[lane]) : 0u;
    o.ownerLane = ownerLane; o.owner = owner; o.c = c; o.summ = summ; o.flags = flags; o.ns = nsCur; o.dropped = owner && dropped; o.dropPos = dropPos;
    FS_WAVE_SYNC();
    return E;
}

// commit: every owner writes its context's list and record word back.  The successors go through the lane's eight words of
// LDS (the hash table's space, free again) to be picked up in their final order.
template <int N> FS_DEV void win_write_back(Coder& m, const WinFetch<N>& f, const WinSolved<N>& o)
{
    const uint32_t lane = (uint32_t)FS_LANE();
    #pragma unroll
    for (int j = 0; j < N; ++j) m.sh->winTab[8u * lane + (uint32_t)j] = f.sc[j];
    FS_WAVE_SYNC();
    // Contexts whose lists lost states in a rescale of this window: their units shrink now (ShrinkUnits, SubAlloc.hpp:170-184 -- a
    // block of the right size from the free list if there is one, else the tail split off), one context after the other in the
    // order of the positions that rescaled them: the free lists are stacks, the order of the calls is part of the model's state.
    // (Nothing else of the window touches the allocator, and the episode behind the window comes behind all of them.)
    uint32_t stats = f.stats;
    for (uint64_t dr = fs_ballot(o.owner && o.dropped); dr != 0ull;) {
        uint32_t j = fs_ctz64(dr), best = FS_UNI(fs_readlane(o.dropPos, j));
        for (uint64_t t = dr & (dr - 1ull); t != 0ull; t &= t - 1ull) { const uint32_t b = fs_ctz64(t), pb = FS_UNI(fs_readlane(o.dropPos, b)); if (pb < best) { best = pb; j = b; } }
        const uint32_t st0 = FS_UNI(fs_readlane(f.stats, j)), ns0 = FS_UNI(fs_readlane(f.ns, j)), ns1 = FS_UNI(fs_readlane(o.ns, j));
        const uint32_t st1 = ShrinkUnits(m, st0, (ns0 + 2u) >> 1, (ns1 + 2u) >> 1);
        if (lane == j) stats = st1;
        dr &= ~(1ull << j);
    }
    if (o.owner) {
        fs_gptr32 p = (fs_gptr32)HP(stats);
        const uint32_t nst = o.ns + 1u;
        uint32_t w[3 * N / 2], sf[N], so[N];
        #pragma unroll
        for (int j = 0; j < N; ++j) {
            sf[j] = ((uint32_t)(o.c.S >> (8 * j)) & 0xFFu) | (((uint32_t)(o.c.F >> (8 * j)) & 0xFFu) << 8);
            so[j] = m.sh->winTab[8u * lane + ((o.c.P >> (4 * j)) & 7u)];
        }
        #pragma unroll
        for (int t = 0; t < N / 2; ++t) {
            w[3 * t] = sf[2 * t] | (so[2 * t] << 16);
            w[3 * t + 1] = (so[2 * t] >> 16) | (sf[2 * t + 1] << 16);
            w[3 * t + 2] = so[2 * t + 1];
        }
        // whole units (two states, three words) go back: the spare half of an odd list's last unit is never read
        const uint32_t units = (nst + 1u) >> 1;
        #pragma unroll
        for (int u = 0; u < N / 2; ++u) if ((uint32_t)u < units) { p[3 * u] = w[3 * u]; p[3 * u + 1] = w[3 * u + 1]; p[3 * u + 2] = w[3 * u + 2]; }
        *(fs_gptr32)HP(f.addr) = o.ns | (o.flags << 8) | (o.summ << 16);
        if (stats != f.stats) *(fs_gptr32)(HP(f.addr) + 4u) = stats;
    }
    FS_WAVE_SYNC();
}

// One window at position `pos` (the serial state is at the top of its loop with OrderFall == 0 and MinContext ==
// MaxContext).  Returns the number of symbols coded, 0 if the first position is not a plain hit.  On return > 0 the model
// memory, the coder, PrevSuccess, MaxContext and `hist` are exactly what the serial walk would have left.
template <int N> FS_DEV uint32_t window_step_n(Coder& m, fs_cgptr in, uint32_t n, uint32_t pos, const WinHead& h, uint32_t& hist, uint64_t tp, const uint64_t tEnter)
{
    WinFetch<N> f; WinSolved<N> o;
    win_fetch_list<N>(m, h, f);
    FS_PROF_ACC(m.sh->winStats[8], tp);                                // input bytes, hint, record, list
    const uint32_t L = win_solve<N>(m, in, n, pos, f, 0u, o, tp);
    if (L == 0u) { uint64_t te = tEnter; FS_PROF_ACC(m.sh->winStats[14], te); return 0u; }
    win_write_back<N>(m, f, o);
    uint32_t tA = o.tA, tM = o.tM;
    if (m.queued) {                                            // two-wave form: the slots go to the coder wave as they are
        cq_push_lanes(m, tA, tM, L);
        FS_PROF_ACC(m.sh->winStats[12], tp);
    } else {
        // every position: its slot, and the reciprocal of its total (all lanes at once)
        if ((uint32_t)FS_LANE() < L) {
            const Recip rc = recip_make(tM);
            tA |= (rc.l - 1u) << 24; tM = rc.mul;
        }
        FS_PROF_ACC(m.sh->winStats[12], tp);                           // write-back, reciprocals

        // the range coder, in stream order (Coder.hpp:13-17 + the normalisation of Model.cpp:580): scalar code
        // (range >= 2^24 implies that low and low + range differ above bit 23: nothing to shift out -- the usual case is
        // decided by one compare; the slots of four symbols are fetched ahead of their chains)
        #define FS_CODE_ONE(A_, M_) do { \
            const uint32_t t_ = fs_mulhi(m.range, M_), rr_ = (t_ + ((m.range - t_) >> 1)) >> (A_ >> 24); \
            m.low += (A_ & 0xFFFFu) * rr_; m.range = rr_ * ((A_ >> 16) & 0x7Fu); \
            if (__builtin_expect(m.range < TOP, 0)) rc_shift_out(m); } while (0)
        uint32_t i_ = 0;
        while (i_ + 4u <= L) {
            const uint32_t A0 = FS_UNI(fs_readlane(tA, i_)), M0 = FS_UNI(fs_readlane(tM, i_)), A1 = FS_UNI(fs_readlane(tA, i_ + 1u)), M1 = FS_UNI(fs_readlane(tM, i_ + 1u));
            const uint32_t A2 = FS_UNI(fs_readlane(tA, i_ + 2u)), M2 = FS_UNI(fs_readlane(tM, i_ + 2u)), A3 = FS_UNI(fs_readlane(tA, i_ + 3u)), M3 = FS_UNI(fs_readlane(tM, i_ + 3u));
            FS_CODE_ONE(A0, M0); FS_CODE_ONE(A1, M1); FS_CODE_ONE(A2, M2); FS_CODE_ONE(A3, M3);
            i_ += 4u;
        }
        for (; i_ < L; ++i_) { const uint32_t A0 = FS_UNI(fs_readlane(tA, i_)), M0 = FS_UNI(fs_readlane(tM, i_)); FS_CODE_ONE(A0, M0); }
        #undef FS_CODE_ONE
    }
    m.PrevSuccess = (FS_UNI(fs_readlane(tA, L - 1u)) >> 23) & 1u;
    m.MaxContext = FS_UNI(fs_readlane(f.succ, L - 1u));
    const uint32_t kl = FS_UNI(fs_readlane(f.key, L - 1u)), sl = FS_UNI(fs_readlane(f.sym, L - 1u));
    hist = (kl >> 8) | (sl << 24);
    FS_STAT_ADD(m.sh->winStats[1], 1u);
    FS_STAT_ADD(m.sh->winStats[2], L);
    FS_PROF_ACC(m.sh->winStats[13], tp);                           // range coder
    { uint64_t te = tEnter; FS_PROF_ACC(m.sh->winStats[14], te); }
    return L;
}
FS_DEV uint32_t window_step(Coder& m, fs_cgptr in, uint32_t n, uint32_t pos, uint32_t MinContext, uint32_t& hist)
{
    uint64_t tp = FS_PROF_NOW(); const uint64_t tEnter = tp;
    WinHead h;
#if defined(FS_HEAP_STATS)
    struct InWin { InWin() { fs_heap_stat_window(1); } ~InWin() { fs_heap_stat_window(0); } } inWin_;
#endif
    win_fetch_head(m, in, n, pos, MinContext, h);
    FS_STAT_ADD(m.sh->winStats[0], 1u);
    return window_step_n<8>(m, in, n, pos, h, hist, tp, tEnter);
}
#undef FS_CE
