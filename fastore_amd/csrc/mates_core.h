// The mate search of ONE paired-end bin by one 1024-thread workgroup (LzCompressorPE::CompressPair's history search,
// FastqCompressor.cpp:4460-4740; the model the kernel implements is spelled out pair by pair in tests/emu/engine_emu.cpp: match_mates).
//
// A bin's pairs come in the order the tree walk emits them and each search depends on where the mates before it went, so a bin is a
// chain of pairs; bins are independent: a launch is a workgroup per bin.  What one pair costs, six barriers in all:
//   * the mate's bases as three bit planes (bits 0, 1, 2 of the base's place in the archive's symbol order): three ballots per 64 bases;
//     a position's signature is a funnel shift of two plane words, a bit reversal and a 256-entry "spread" table -- a thread per
//     position, no loop over the signature's bases; a window that holds an 'N' has a bit of plane 2;
//   * the two sets (distinct valid signatures of the first part; of the second part and not in the first) live in ONE open-addressing
//     table in LDS, ((set << 24 | signature << 8 | first position) + 1): compare-and-swap claims a slot, a minimum keeps the first
//     position, so what the table holds does not depend on which thread came first;
//   * the history -- at most 1024 mates, a THREAD each -- keeps its listed signatures in LDS and its planes in the bin's scratch in global
//     memory (L2-resident); an entry that lists a member signature is a candidate = four ALIGNMENTS (the entry's four stored positions
//     against the signature's first position in the mate, unused zeros included, as the reference does);
//   * an alignment is priced by one thread: the entry's planes against the mate's, 32 bases per exclusive-or.  In a bin of a deep library
//     nearly EVERY history entry is a candidate (measured: 1 116 candidates, 4 500 alignments a pair), so what is priced is bounded: an
//     alignment costs at least |shift| x s, and one whose bound is ABOVE the cheapest cost found cannot win (at the same cost it still can:
//     the order among equals decides).  The first pass lists the alignments whose bound is within a guess (the pair before's cost + 4);
//     only if the cheapest cost ends up above the guess a second pass, a quarter of the history at a time, lists what lies between;
//   * the cheapest by (cost, signature, age: oldest first, stored position) -- the reference's "first among equals" -- through a
//     64-bit LDS minimum.
// The history is a ring: a mate that matched at cost 0 "goes to the back and is dropped by the next pair" -- it is simply not
// pushed; the slot the ring points at is the entry the next push overwrites, it left the history when the history became full.
// An entry's age follows from its slot and the number of pushes, so entries carry neither stamps nor "live" flags.
// (Rounds 3 / 4, measured on the way here: a wavefront per alignment with a lane per base and every candidate's four alignments looked
// at by one wavefront after the other, ~20 us a pair; ONE wavefront per bin, 64 entries a step -- a lone wavefront has nothing to hide
// its LDS round trips and exec-mask arithmetic behind: 45 us a pair, profiles/r04_mate_search_forms.txt.)
//
// The same source runs on the lock-step wave emulation of tests/emu/simt.h, sixteen emulated wavefronts and a polled barrier
// (tests/test_simt.py holds its rows against the host's search on every golden paired-end bin); nothing here is compiled into a host
// path of the product.
#pragma once
#include "wave.h"
#include "device_types.h"

namespace fsmate {
using namespace fsdev;

// (FSM_ITEMS: a shorter list for the tests, so that the passes over a part of the history at a time are met on small bins too)
#if !defined(FSM_ITEMS)
  #define FSM_ITEMS 4096
#endif
// (FSM_THREADS: the workgroup's size, 256 .. 1024: a thread takes kWindowMax / kThreads history entries)
#if !defined(FSM_THREADS)
  #define FSM_THREADS 1024
#endif
enum : uint32_t { kThreads = FSM_THREADS, kWaves = kThreads / 64u, kEPT = 1024u / kThreads, kWindowMax = 1024, kHash = 1024, kHashShift = 22, kItems = FSM_ITEMS, kPartEntries = kItems / 16u /* sixteen alignments an entry at most: a part's never overflow the list */,
                  kParts = (kWindowMax + kPartEntries - 1u) / kPartEntries, kNone = 0xFFFFFFFFu, kPlaneWords = 9,
                  kEntryWords = 4u + 4u * kPlaneWords };

struct alignas(16) Shared {
    uint32_t hash[kHash];                 // the mate's member signatures
    uint32_t planes[kPlaneWords][4];      // the mate's planes, a word of each per 32 bases (+ one word of zeros behind)
    uint32_t ring4[kWindowMax][4];        // the history's listed signatures, a slot each: signature | position << 16 (word 0: | length << 24); a word 0: unused
    uint32_t items[kItems];               // alignments to price: entry slot << 12 | listed signature's index << 10 | stored position's index << 8 | the signature's first position in the mate
    uint32_t sigAt[256];                  // per position of the mate: its signature | set << 16 if the position counts for a set, else kNone
    uint32_t small1[4], small2[4];        // the sets' smallest members: signature << 8 | first position
    uint32_t pairOf[kWindowMax];          // the pair whose mate a slot holds
    uint32_t size1, size2, nItems, overflow, nAll;
    unsigned long long best;              // (cost << 16 | signature) << 32 | order among equals
    uint16_t spread[256];                 // bit i of the index at bit 2 i
    uint32_t arrived;                     // (the emulation's barrier)
};

// a bin's scratch in global memory (32-bit words): per history slot four words of nothing and its planes
FS_DEV uint32_t hist_words(uint32_t window) { return window * kEntryWords; }

#if FS_WIDE      // the kernel's body: the device build and the lock-step test emulation

#if defined(__HIP_DEVICE_COMPILE__)
  #define FSM_CAS(w, e, v) atomicCAS(&(w), (e), (v))
  #define FSM_MIN(w, v) ((void)atomicMin(&(w), (v)))
  #define FSM_MIN64(w, v) ((void)atomicMin(&(w), (v)))
  #define FSM_INC(w) atomicAdd(&(w), 1u)
  #define FSM_ADD(w, x) atomicAdd(&(w), (x))
  FS_DEV uint32_t rev8(uint32_t x) { return __builtin_bitreverse32(x) >> 24; }
  FS_DEV uint32_t popc32(uint32_t x) { return (uint32_t)__popc(x); }
  #define FSM_WAVE() ((uint32_t)threadIdx.x >> 6)
  // the workgroup's barrier; the history's planes in global memory are written and read by threads of this workgroup only (one compute
  // unit, one vector cache), so the barrier's workgroup-scope ordering is all they need
  #define FSM_SYNC(sh, gen) __syncthreads()
#else
  static inline uint32_t fsm_cas(uint32_t& w, uint32_t e, uint32_t v) { const uint32_t o = w; if (o == e) w = v; return o; }
  #define FSM_CAS(w, e, v) fsm_cas((w), (e), (v))
  #define FSM_MIN(w, v) do { if ((v) < (w)) (w) = (v); } while (0)
  #define FSM_MIN64(w, v) do { if ((v) < (w)) (w) = (v); } while (0)
  static inline uint32_t fsm_inc(uint32_t& w) { return w++; }
  #define FSM_INC(w) fsm_inc(w)
  static inline uint32_t fsm_add(uint32_t& w, uint32_t x) { const uint32_t o = w; w += x; return o; }
  #define FSM_ADD(w, x) fsm_add((w), (x))
  static inline uint32_t rev8(uint32_t x) { uint32_t r = 0; for (int i = 0; i < 8; ++i) r |= ((x >> i) & 1u) << (7 - i); return r; }
  static inline uint32_t popc32(uint32_t x) { return (uint32_t)__builtin_popcount(x); }
  #define FSM_WAVE() ((uint32_t)simt::wave())
  // sixteen emulated wavefronts meet: each one's lanes meet, one lane reports, all poll (a poll is a meeting point, so the other
  // wavefronts get their turns)
  #define FSM_SYNC(sh, gen) do { simt::barrier(); ++(gen); if (simt::lane() == 0) ++(sh).arrived; while ((sh).arrived < (gen) * kWaves) simt::barrier(); simt::barrier(); } while (0)
#endif

FS_DEV uint32_t wave_min(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // DPP min scan: row_shr 1/2/4/8 inside each row of 16, row_bcast 15 / 31 across the rows; the wave's minimum ends in lane 63
    int v = (int)x;
    #define FSM_MIN_STEP(ctrl, rowmask) do { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp((int)kNone, v, ctrl, rowmask, 0xf, false); \
                                             v = (int)(((uint32_t)v < o_) ? (uint32_t)v : o_); } while (0)
    FSM_MIN_STEP(0x111, 0xf); FSM_MIN_STEP(0x112, 0xf); FSM_MIN_STEP(0x114, 0xf); FSM_MIN_STEP(0x118, 0xf);
    FSM_MIN_STEP(0x142, 0xa); FSM_MIN_STEP(0x143, 0xc);
    #undef FSM_MIN_STEP
    return (uint32_t)__builtin_amdgcn_readlane(v, 63);
#else
    for (uint32_t m = 1; m < 64u; m <<= 1) { const uint32_t o = fs_bperm(x, (uint32_t)FS_LANE() ^ m); x = o < x ? o : x; }
    return x;
#endif
}

// bits [s, s + 32) of hi:lo
FS_DEV uint32_t funnel(uint32_t lo, uint32_t hi, uint32_t s) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (s & 31u)); }

struct alignas(16) Quad { uint32_t x, y, z, w; };
FS_DEV Quad ldq(const uint32_t* p) { return *(const Quad*)p; }      // (16-byte aligned: one 128-bit load)
FS_DEV uint32_t sel4(const Quad& q, uint32_t i) { return i == 0u ? q.x : (i == 1u ? q.y : (i == 2u ? q.z : q.w)); }
FS_DEV uint32_t sel4(const uint32_t (&a)[4], uint32_t i) { return i == 0u ? a[0] : (i == 1u ? a[1] : (i == 2u ? a[2] : a[3])); }
#if defined(__HIP_DEVICE_COMPILE__)
  #define FSM_UNROLL _Pragma("unroll")
#else
  #define FSM_UNROLL
#endif

// phase clocks of a bin's chain (a -DFSM_PROFILE build of matcher.hip prints them per bin; design studies)
#if defined(FSM_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
  #define FSM_T(i) do { const uint64_t n_ = __builtin_amdgcn_s_memtime(); prof[i] += n_ - tprof; tprof = n_; } while (0)
#else
  #define FSM_T(i) ((void)0)
#endif
// design-study counters (tests/emu/mates_simt.cpp with -DFSM_COUNT); nothing otherwise
#if !defined(FSM_STAT)
  #define FSM_STAT(i, x) ((void)0)
#endif

// the place of the entry in slot `s` in the history, newest = 0 (ring: the slot the next push goes to)
FS_DEV uint32_t age_of(uint32_t s, uint32_t ring, uint32_t W) { return s > ring ? s - ring - 1u : s + W - ring - 1u; }

// `hist`: hist_words(par.window) words of global scratch of this bin alone (no initial contents assumed, nor of `sh`)
FS_DEV void search_bin(Shared& sh, const MateJob job, const MatePair* pairs, const uint8_t* seq, const uint32_t* validBits, const MateParams& par, MateRow* rows, uint32_t* hist)
{
    const uint32_t lane = (uint32_t)FS_LANE(), wave = FSM_WAVE(), tid = wave * 64u + lane;
    const uint32_t W = par.window, L = par.sig_len, sigBits = (1u << L) - 1u;
    const uint32_t sc = (uint32_t)par.shift_cost, mc = (uint32_t)par.mismatch_cost;
    const uint32_t inv20 = sc != 0u ? ((1u << 20) - 1u) / sc + 1u : 0u;      // x / s == (x * inv20) >> 20 for the x met here (below 4 096)
    uint32_t gen = 0; (void)gen;
    if (tid < 256u) { uint32_t v = 0; for (uint32_t b = 0; b < 8u; ++b) v |= ((tid >> b) & 1u) << (2u * b); sh.spread[tid] = (uint16_t)v; }
    if (tid < 4u) sh.planes[kPlaneWords - 1u][tid] = 0u;
    if (tid == 0u) sh.arrived = 0u;
    uint32_t ring = W - 1u, pushes = 0u, guess = 8u;
    bool dense = false;                     // the pair before met more alignments than the list holds: list within the guess first
    if (job.count == 0u) return;
    // (a pair's bases are asked for one pair ahead: the first touch of them comes from HBM)
    MatePair prNext = pairs[job.first];
    uint32_t baseNext = wave < 4u && tid < prNext.mate_len ? (uint32_t)seq[prNext.mate_off + tid] : 0u;
#if defined(FSM_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprof = __builtin_amdgcn_s_memtime();
#endif
    for (uint32_t p = 0; p < job.count; ++p) {
        const MatePair pr = prNext;
        const uint32_t plen = pr.mate_len, base = baseNext;
        if (p + 1u < job.count) { prNext = pairs[job.first + p + 1u]; baseNext = wave < 4u && tid < prNext.mate_len ? (uint32_t)seq[prNext.mate_off + tid] : 0u; }
        // ---- the mate's planes (wavefronts 0..3: 64 positions each); the table and the pair's counters start empty
        for (uint32_t i = tid; i < kHash; i += kThreads) sh.hash[i] = 0u;
        if (tid == 0u) { sh.nItems = 0u; sh.size1 = 0u; sh.size2 = 0u; sh.overflow = 0u; sh.nAll = 0u; sh.best = ~0ull; }
        if (wave < 4u) {
            uint32_t code = 4u;
            if (tid < plen) { for (uint32_t k = 0; k < 4u; ++k) if (par.symbol_order[k] == base) code = k; }
            const uint64_t m0 = fs_ballot((code & 1u) != 0u), m1 = fs_ballot((code & 2u) != 0u), m2 = fs_ballot((code & 4u) != 0u);
            if (lane < 2u) {
                uint32_t* w = sh.planes[2u * wave + lane];
                w[0] = lane ? (uint32_t)(m0 >> 32) : (uint32_t)m0; w[1] = lane ? (uint32_t)(m1 >> 32) : (uint32_t)m1; w[2] = lane ? (uint32_t)(m2 >> 32) : (uint32_t)m2; w[3] = 0u;
            }
        }
        FSM_SYNC(sh, gen);
        FSM_T(0);
        // ---- the signatures of its positions (FindMinimizers over the two parts), a thread each; the two sets: first part, then what the second adds
        const int32_t half = (int32_t)plen / 2;
        const int32_t end1 = (int32_t)plen - (int32_t)L - ((int32_t)par.skip_zone + half - ((int32_t)L - 1));
        const int32_t end2 = (int32_t)plen - (int32_t)L - (int32_t)par.skip_zone;
        uint32_t sig = 0u, kind = 0u;              // kind: 0 nothing, 1 a position of the first part, 2 of the second part only
        if (tid < 256u) {
            const bool in1 = (int32_t)tid < end1, in2 = (int32_t)tid >= half && (int32_t)tid < end2;
            if (tid + L <= plen && (in1 || in2)) {
                const uint32_t i = tid >> 5, s = tid & 31u;
                const Quad lo = ldq(sh.planes[i]), hi = ldq(sh.planes[i + 1u]);
                const uint32_t x0 = funnel(lo.x, hi.x, s) & sigBits, x1 = funnel(lo.y, hi.y, s) & sigBits, x2 = funnel(lo.z, hi.z, s) & sigBits;
                if (x2 == 0u) {
                    const uint32_t m = ((uint32_t)sh.spread[rev8(x1) >> (8u - L)] << 1) | (uint32_t)sh.spread[rev8(x0) >> (8u - L)];
                    if ((validBits[m >> 5] >> (m & 31u)) & 1u) { sig = m; kind = in1 ? 1u : 2u; }
                }
            }
        }
        uint32_t member = kNone;                   // signature | set << 16 when this position counts for a set
        for (uint32_t set = 0; set < 2u; ++set) {
            if (kind == set + 1u) {
                const uint32_t v = ((set << 24) | (sig << 8) | tid) + 1u;
                for (uint32_t h = (sig * 0x9E3779B1u) >> kHashShift;; h = (h + 1u) & (kHash - 1u)) {
                    const uint32_t old = FSM_CAS(sh.hash[h], 0u, v);
                    if (old == 0u) { (void)FSM_INC(set ? sh.size2 : sh.size1); member = sig | (set << 16); break; }
                    if ((((old - 1u) >> 8) & 0xFFFFu) == sig) { if (((old - 1u) >> 24) == set) { FSM_MIN(sh.hash[h], v); member = sig | (set << 16); } break; }      // (in the first set already: not a member of the second)
                }
            }
            if (set == 1u && tid < 256u) sh.sigAt[tid] = member;
            FSM_SYNC(sh, gen);
        }
        FSM_T(1);
        // ---- the sets' four smallest members with their first positions (wavefront 0: set 1, wavefront 1: set 2) ...
        if (wave < 2u) {
            uint32_t key[4];
            FSM_UNROLL for (uint32_t q = 0; q < 4u; ++q) {
                const uint32_t t = lane + 64u * q, sg = sh.sigAt[t];
                key[q] = sg != kNone && (sg >> 16) == wave ? ((sg & 0xFFFFu) << 8) | t : kNone;
            }
            FSM_UNROLL for (uint32_t r = 0; r < 4u; ++r) {
                uint32_t loc = key[0] < key[1] ? key[0] : key[1]; const uint32_t l2 = key[2] < key[3] ? key[2] : key[3]; loc = loc < l2 ? loc : l2;
                const uint32_t best = wave_min(loc);
                if (lane == 0u) (wave == 0u ? sh.small1 : sh.small2)[r] = best;
                FSM_UNROLL for (uint32_t q = 0; q < 4u; ++q) if (best != kNone && (key[q] >> 8) == (best >> 8)) key[q] = kNone;
            }
        }
        // ---- ... and the history, a thread per entry: its listed signatures against the table; the alignments whose bound |shift| x s lies in
        // [low, lim] go on the list (entries [eLo, eHi))
        const uint32_t lowValid = pushes >= W ? 0u : W - pushes;
        Quad e4s[kEPT]; bool lives[kEPT];
        FSM_UNROLL for (uint32_t u = 0; u < kEPT; ++u) {
            const uint32_t e = tid + kThreads * u;
            lives[u] = e < W && e >= lowValid && !(pushes >= W && e == ring);
            e4s[u].x = e4s[u].y = e4s[u].z = e4s[u].w = 0u;
            if (lives[u]) e4s[u] = ldq(sh.ring4[e]);
        }
        auto list_one = [&](uint32_t ent, const Quad& e4, bool live, uint32_t low, uint32_t lim, uint32_t eLo, uint32_t eHi) {
            const uint32_t minShift = sc != 0u ? ((low + sc - 1u) * inv20) >> 20 : (low != 0u ? 128u : 0u);         // |shift| x s >= low
            const uint32_t maxShift = sc != 0u ? (lim * inv20) >> 20 : 127u;                                          // |shift| x s <= lim
            uint32_t want = 0u, posOf = 0u, all = 0u;      // bit 4 j + k: alignment (j, k) goes on the list; byte j: the signature's first position in the mate; all: alignments there are
            if (live && ent >= eLo && ent < eHi) {
                uint32_t first[4];
                FSM_UNROLL for (uint32_t j = 0; j < 4u; ++j) { const uint32_t sj = sel4(e4, j) & 0xFFFFu; first[j] = sj != 0u ? sh.hash[(sj * 0x9E3779B1u) >> kHashShift] : 0u; }
                FSM_UNROLL for (uint32_t j = 0; j < 4u; ++j) {
                    const uint32_t sj = sel4(e4, j) & 0xFFFFu;
                    uint32_t v = first[j];
                    if (v != 0u && (((v - 1u) >> 8) & 0xFFFFu) != sj)          // (another signature's slot: on along the probe sequence)
                        for (uint32_t h = ((sj * 0x9E3779B1u) >> kHashShift) + 1u;; ++h) { v = sh.hash[h & (kHash - 1u)]; if (v == 0u || (((v - 1u) >> 8) & 0xFFFFu) == sj) break; }
                    if (v == 0u) continue;
                    const uint32_t posH = (v - 1u) & 0xFFu;
                    posOf |= posH << (8u * j);
                    FSM_UNROLL for (uint32_t k = 0; k < 4u; ++k) {
                        const int32_t shift = (int32_t)((sel4(e4, k) >> 16) & 0xFFu) - (int32_t)posH;
                        const uint32_t ashift = (uint32_t)(shift < 0 ? -shift : shift);
                        if (ashift <= 127u) { ++all; if (ashift <= maxShift && ashift >= minShift) want |= 1u << (4u * j + k); }
                    }
                }
            }
            // a wavefront's alignments take their places on the list together: one addition to the workgroup's counter per wavefront
            const uint32_t mine = popc32(want);
            const uint32_t cnt = (mine << 16) | all;                       // (both sums in one pass: at most 64 x 16 each)
            uint32_t run = cnt;
#if defined(__HIP_DEVICE_COMPILE__)
            { int x = (int)run;
              x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true); x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
              x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true); x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
              x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
              run = (uint32_t)x; }
#else
            for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t o = fs_bperm(run, lane >= d ? lane - d : lane); if (lane >= d) run += o; }
#endif
            const uint32_t total = fs_readlane(run, 63u);
            if (total == 0u) return;
            uint32_t at0 = 0u;
            if (lane == 63u) { if (total >> 16) at0 = FSM_ADD(sh.nItems, total >> 16); if (total & 0xFFFFu) (void)FSM_ADD(sh.nAll, total & 0xFFFFu); }
            at0 = fs_readlane(at0, 63u);
            uint32_t at = at0 + (run >> 16) - mine;
            for (uint32_t m = want; m != 0u; m &= m - 1u, ++at) {
                const uint32_t bit = (uint32_t)__builtin_ctz(m), j = bit >> 2, k = bit & 3u;
                if (at < kItems) sh.items[at] = (ent << 12) | (j << 10) | (k << 8) | ((posOf >> (8u * j)) & 0xFFu); else sh.overflow = 1u;
            }
        };
        auto list = [&](uint32_t low, uint32_t lim, uint32_t eLo, uint32_t eHi) {
            FSM_UNROLL for (uint32_t u = 0; u < kEPT; ++u) list_one(tid + kThreads * u, e4s[u], lives[u], low, lim, eLo, eHi);
        };
        // what is on the list, a thread per alignment; the cheapest goes to the workgroup's minimum
        auto price = [&](uint32_t bound) {
            const uint32_t n = sh.nItems < kItems ? sh.nItems : (uint32_t)kItems;
            uint32_t hi = kNone, lo = kNone;
            for (uint32_t it = tid; it < n; it += kThreads) {
                const uint32_t w = sh.items[it], s = w >> 12, j = (w >> 10) & 3u, k = (w >> 8) & 3u, posH = w & 0xFFu;
                const Quad f4 = ldq(sh.ring4[s]);
                const uint32_t ek = sel4(f4, k), ej = sel4(f4, j);
                const int32_t shift = (int32_t)((ek >> 16) & 0xFFu) - (int32_t)posH;
                const uint32_t ashift = (uint32_t)(shift < 0 ? -shift : shift);
                const uint32_t elen = f4.x >> 24;
                const uint32_t recOff = shift < 0 ? ashift : 0u, lzOff = shift > 0 ? ashift : 0u;
                const uint32_t a = plen - recOff, b = elen - lzOff, minLen = a < b ? a : b;
                const uint32_t* ep = hist + kEntryWords * s + 4u;
                uint32_t mw = recOff >> 5, ew = lzOff >> 5; const uint32_t ms = recOff & 31u, es = lzOff & 31u;
                uint32_t mism = 0;
                // (the first six words of both sides are asked for at once -- 160 bases, one round trip to the L2 --, longer overlaps go on word by word)
                Quad mq[6], eq[6];
                FSM_UNROLL for (uint32_t i = 0; i < 6u; ++i) { const uint32_t a1 = mw + i, b1 = ew + i; mq[i] = ldq(sh.planes[a1 < kPlaneWords - 1u ? a1 : kPlaneWords - 1u]); eq[i] = ldq(ep + 4u * (b1 < kPlaneWords - 1u ? b1 : kPlaneWords - 1u)); }
                FSM_UNROLL for (uint32_t i = 0; i < 5u; ++i) {
                    if (32u * i < minLen) {
                        uint32_t d = (funnel(mq[i].x, mq[i + 1u].x, ms) ^ funnel(eq[i].x, eq[i + 1u].x, es)) | (funnel(mq[i].y, mq[i + 1u].y, ms) ^ funnel(eq[i].y, eq[i + 1u].y, es)) |
                                     (funnel(mq[i].z, mq[i + 1u].z, ms) ^ funnel(eq[i].z, eq[i + 1u].z, es));
                        const uint32_t rem = minLen - 32u * i;
                        if (rem < 32u) d &= (1u << rem) - 1u;
                        mism += popc32(d);
                    }
                }
                if (minLen > 160u) {
                    Quad mlo = mq[5], elo = eq[5];
                    mw += 5u; ew += 5u;
                    for (uint32_t done = 160u; done < minLen; done += 32u) {
                        ++mw; ++ew;
                        const Quad mhi = ldq(sh.planes[mw < kPlaneWords - 1u ? mw : kPlaneWords - 1u]), ehi = ldq(ep + 4u * (ew < kPlaneWords - 1u ? ew : kPlaneWords - 1u));
                        uint32_t d = (funnel(mlo.x, mhi.x, ms) ^ funnel(elo.x, ehi.x, es)) | (funnel(mlo.y, mhi.y, ms) ^ funnel(elo.y, ehi.y, es)) | (funnel(mlo.z, mhi.z, ms) ^ funnel(elo.z, ehi.z, es));
                        const uint32_t rem = minLen - done;
                        if (rem < 32u) d &= (1u << rem) - 1u;
                        mism += popc32(d);
                        mlo = mhi; elo = ehi;
                    }
                }
                const uint32_t cost = ashift * sc + mism * mc;
                if (cost <= bound) {
                    const uint32_t h2 = (cost << 16) | (ej & 0xFFFFu), l2 = ((W - 1u - age_of(s, ring, W)) << 22) | (k << 20) | (s << 10) | (posH << 2);
                    if (h2 < hi || (h2 == hi && l2 < lo)) { hi = h2; lo = l2; }
                }
            }
            if (n != 0u) {
                const uint32_t mh = wave_min(hi);
                const uint32_t ml = wave_min(hi == mh ? lo : kNone);
                if (lane == 0u && mh != kNone) FSM_MIN64(sh.best, ((unsigned long long)mh << 32) | ml);
            }
        };
        // first pass: every alignment there is -- unless the pair before met more than the list holds: then those within the guess
        const uint32_t lim1 = dense ? (guess < 254u ? guess : 254u) : 254u;
        list(0u, lim1, 0u, kWindowMax);
        FSM_SYNC(sh, gen);
        FSM_T(2);
        price(254u);
        FSM_SYNC(sh, gen);
        FSM_T(3);
        dense = sh.nAll > kItems;
        // (the list was too short, or the guess too low: what lies between the guess and the cheapest cost found -- all of the history at once
        // if that fits the list, else a quarter of it at a time)
        bool over = sh.overflow != 0u;
        if (over || (lim1 < 254u && (sh.best == ~0ull || (uint32_t)(sh.best >> 48) > lim1))) {
            uint32_t low = over ? 0u : lim1 + 1u;
            for (uint32_t part = 0; part <= kParts; ++part) {             // part 0: everything; parts 1..: kPartEntries entries at a time (quarters in the product), when part 0 did not fit
                if (part == 1u && !over) break;
                const uint32_t bound = sh.best == ~0ull ? 254u : (uint32_t)(sh.best >> 48);
                FSM_SYNC(sh, gen);                                      // (everyone has read the list's state of the pass before)
                if (tid == 0u) { sh.nItems = 0u; sh.overflow = 0u; }
                FSM_SYNC(sh, gen);
                if (bound >= low) list(low, bound, part == 0u ? 0u : (part - 1u) * kPartEntries, part == 0u ? (uint32_t)kWindowMax : part * kPartEntries);
                FSM_SYNC(sh, gen);
                if (part == 0u) { over = sh.overflow != 0u; if (over) continue; }      // (did not fit: nothing of it is priced, the quarters do it all)
                price(bound);
                FSM_SYNC(sh, gen);
            }
        }
        FSM_T(4);
        // ---- the answer
        const unsigned long long best = sh.best;
        uint32_t cost = 255u, prevId = 0, matchPair = 0; int32_t shift = 0; bool noMism = false;
        if (best != ~0ull) {
            const uint32_t bestLo = (uint32_t)best;
            cost = (uint32_t)(best >> 48);
            const uint32_t s = (bestLo >> 10) & 1023u, k = (bestLo >> 20) & 3u, posH = (bestLo >> 2) & 0xFFu;
            shift = (int32_t)((sh.ring4[s][k] >> 16) & 0xFFu) - (int32_t)posH;
            prevId = age_of(s, ring, W); matchPair = sh.pairOf[s];
            const uint32_t ashift = (uint32_t)(shift < 0 ? -shift : shift);
            noMism = cost == ashift * sc;
        }
        FSM_STAT(0, 1);
        const bool matched = (int32_t)cost <= (int32_t)pr.threshold;
        const bool identical = matched && noMism && cost == 0u;
        if (tid == 0u) {
            MateRow row; row.match = matched ? (int32_t)matchPair : -1; row.cost = (int16_t)cost; row.shift = (int16_t)shift; row.prev_id = (uint16_t)prevId;
            row.no_mismatches = noMism ? 1 : 0; row.overflow = 0;
            rows[job.first + p] = row;
        }
        if (best != ~0ull) { const uint32_t c = cost + 4u; guess = c < 8u ? 8u : (c > 32u ? 32u : c); }
        // ---- the mate's own entry, into the slot the ring points at: nobody reads that slot during this pair, and it only counts once the ring
        // moves on -- which it does unless the mate "went to the back" (then the next pair writes over it)
        if (tid < 4u) {
            // two signatures from the smaller set, then from the other one up to four in all
            const uint32_t size1 = sh.size1, size2 = sh.size2;
            const bool swap = size1 > size2;
            const uint32_t firstSize = swap ? size2 : size1, total = size1 + size2;
            const uint32_t nFirst = firstSize < 2u ? firstSize : 2u, nAll = total < 4u ? total : 4u;
            uint32_t mine = 0u;
            if (tid < nFirst) { const uint32_t x = swap ? sh.small2[tid] : sh.small1[tid]; mine = (x >> 8) | ((x & 0xFFu) << 16); }
            else if (tid < nAll) { const uint32_t r = tid - nFirst; const uint32_t x = swap ? sh.small1[r] : sh.small2[r]; mine = (x >> 8) | ((x & 0xFFu) << 16); }
            sh.ring4[ring][tid] = tid == 0u ? mine | (plen << 24) : mine;
            if (tid == 0u) sh.pairOf[ring] = p;
        }
        if (tid < 4u * kPlaneWords) hist[kEntryWords * ring + 4u + tid] = sh.planes[tid >> 2][tid & 3u];
        FSM_SYNC(sh, gen);                                                  // (everyone has read the pair's state: the next pair may begin to write its own)
        if (!identical) { ring = ring == 0u ? W - 1u : ring - 1u; ++pushes; }
        FSM_T(5);
    }
#if defined(FSM_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    if (tid == 0u && job.count >= 2000u)
        printf("[mates] %u pairs: clocks per pair: planes %llu, signatures + sets %llu, smallest + listing %llu, pricing %llu, second pass %llu, answer + entry %llu\n", job.count,
               (unsigned long long)(prof[0] / job.count), (unsigned long long)(prof[1] / job.count), (unsigned long long)(prof[2] / job.count), (unsigned long long)(prof[3] / job.count),
               (unsigned long long)(prof[4] / job.count), (unsigned long long)(prof[5] / job.count));
#endif
}

#endif  // FS_WIDE

}  // namespace fsmate
