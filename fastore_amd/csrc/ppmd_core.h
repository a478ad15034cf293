// PPMd var.J (PPMII) order-4 encoder, one stream per wavefront  (SURVEY §8 a12).
//
// Produces the exact member bytes of the reference path
//   PpmdEncoder::EncodeNextMember -> ppmd_compress -> EncodeFile
//   (/root/reference/fastore/ppmd/PPMd.cpp:119-154, Model.cpp:559-586; model Model.cpp:109-540;
//    sub-allocator SubAlloc.hpp:65-199; carry-less range coder Coder.hpp:7-28)
// with the parameters fastore_pack uses (order 4, 16 MiB sub-allocator, no cut-off:
// fastore_pack/FastqCompressor.h:107-108, FastqCompressor.cpp:772-774).
//
// MI355X mapping: the model heap is a private 16 MiB arena in HBM addressed by 32-bit indices
// (index = byte offset + 1, 0 = null -- the numeric values the reference keeps in its
// iSuccessor/iStats/iSuffix fields, so "successor >= UnitsStart" style tests are plain integer
// compares); the adaptive side tables (binary SEE, SEE2, symbol mask) live in LDS; the coder
// registers stay in SGPR/VGPRs of the wave.  Control flow is wave-uniform (wave.h); table
// initialisation, mask clears and unit copies are spread over the 64 lanes.
#pragma once
#include "wave.h"

namespace fsppmd {

enum { UNIT_SIZE = 12, N1 = 4, N2 = 4, N3 = 4, N4 = (128 + 3 - 1 * N1 - 2 * N2 - 3 * N3) / 4, N_INDEXES = N1 + N2 + N3 + N4 };
enum { INT_BITS = 7, PERIOD_BITS = 7, TOT_BITS = INT_BITS + PERIOD_BITS, INTERVAL = 1 << INT_BITS,
       BIN_SCALE = 1 << TOT_BITS, ROUND = 16, MAX_FREQ = 124 };
enum : uint32_t { TOP = 1u << 24, BOT = 1u << 15 };
constexpr int32_t INIT_RL = -4;      // InitRL of an order-4 model (the value StartModelRare computes from MaxOrder)
enum : uint32_t { SA_SIZE = 16u << 20, MAX_ORDER = 4,
                  // list heads BList[0..N_INDEXES] + one scratch head live behind the heap proper
                  HEADS_OFF = SA_SIZE + 64u,
                  // hint table of the windowed hit path (ppmd_window.h): 2^16 entries {last four bytes, context index}
                  HINT_OFF = (HEADS_OFF + 12u * (N_INDEXES + 2) + 16u + 63u) & ~63u, ARENA_BYTES = HINT_OFF + (8u << 16) };

#if defined(__HIP_DEVICE_COMPILE__)
  #define FS_TABLE __constant__ static const
#else
  #define FS_TABLE static const
#endif
// constants of PPMD_STARTUP (Model.cpp:88-108), tabulated
FS_TABLE uint8_t kIndx2Units[N_INDEXES] = {1, 2, 3, 4, 6, 8, 10, 12, 15, 18, 21, 24, 28, 32, 36, 40, 44, 48, 52, 56,
                                           60, 64, 68, 72, 76, 80, 84, 88, 92, 96, 100, 104, 108, 112, 116, 120, 124, 128};
FS_TABLE uint8_t kUnits2Indx[128] = {
    0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 8, 9, 9, 9, 10, 10, 10, 11, 11, 11, 12, 12, 12, 12, 13, 13, 13, 13,
    14, 14, 14, 14, 15, 15, 15, 15, 16, 16, 16, 16, 17, 17, 17, 17, 18, 18, 18, 18, 19, 19, 19, 19, 20, 20, 20, 20,
    21, 21, 21, 21, 22, 22, 22, 22, 23, 23, 23, 23, 24, 24, 24, 24, 25, 25, 25, 25, 26, 26, 26, 26, 27, 27, 27, 27,
    28, 28, 28, 28, 29, 29, 29, 29, 30, 30, 30, 30, 31, 31, 31, 31, 32, 32, 32, 32, 33, 33, 33, 33, 34, 34, 34, 34,
    35, 35, 35, 35, 36, 36, 36, 36, 37, 37, 37, 37};
FS_TABLE uint8_t kExpEscape[16] = {51, 43, 18, 12, 11, 9, 8, 7, 6, 5, 4, 3, 3, 2, 2, 2};
FS_TABLE int8_t kEscCoef[12] = {16, -10, 1, 51, 14, 89, 23, 35, 64, 26, -42, 43};

FS_DEV uint32_t NS2BSIndx(uint32_t ns) { return ns == 0 ? 0u : (ns < 3 ? 2u : (ns < 29 ? 4u : 6u)); }
// QTable[i] = i for i < 5; then runs of length 1,2,3,... of 5,6,7,...  (Model.cpp:103-107)
FS_DEV uint32_t QTable(uint32_t i)
{
    if (i < 5) return i;
    uint32_t j = i - 5, m = 5, step = 1;        // closed form of the run-length table
    // j < 1 -> 5 ; j < 1+2 -> 6 ; j < 1+2+3 -> 7 ...   (i <= 259 => m <= 26)
    while (j >= step) { j -= step; ++step; ++m; }
    return m;
}

// side tables: LDS on the device (one instance per wave)
struct Shared {
    uint16_t BinSumm[25 * 64];
    uint32_t SEE2[23 * 32];      // Summ | Shift << 16 | Count << 24
    uint8_t CharMask[256];
    uint8_t QT[260];             // QTable, tabulated once per wave
    uint32_t GlueCount, GlueCount1, restarts;   // touched only by the allocator's rare paths / model restarts: kept out of the registers
    // the sub-allocator's free-list heads (BList[0..N_INDEXES] + one scratch head) of the two-wave form: first block of each
    // list, 0 = empty (blk_head).  The heads' Stamp counters (SubAlloc.hpp:41-55) are never read by var.J's allocator and are not kept.
    uint32_t blHead[N_INDEXES + 2];
    uint32_t winA[128], winM[128], winCut;      // [64..127]: spare slots for lanes that have nothing to store in a round
    uint32_t winTab[512], winMask[128];      // owner search: lowest lane per hash slot, per-owner position masks; then the successor words on their way back      // windowed hit path: per-position results, first position that must go back to the serial path
    // coder queue (two-wave form: the model wave produces, the coder wave consumes; see the range-coder section)
    uint32_t qA[256], qM[256];
    uint32_t qTail, qHead;
    // Stream mailboxes: output buffer, capacity and where the size goes, for the stream whose CQ_START is in the ring.  Two
    // of them, used alternately: the model wave may only rewrite box (s & 1) for its stream number s once the coder wave has
    // READ that box for stream s - 2, which qStarts (streams the coder wave has opened) tells it -- qHead alone does not:
    // the coder wave frees a batch's ring slots when the batch is in its registers, before it has worked the batch off.
    uint32_t qBox[2][6];          // qOutLo, qOutHi, qOutCap, qSizeLo, qSizeHi, the stream's issue priority (range-coded and QVZ streams: their pass is the coder wave's)
    uint32_t qStarts;             // written by the coder wave
    uint32_t qOpened;             // model wave only: streams it has started in this workgroup
    // (LDS per workgroup stays below 12 800 bytes: above it a compute unit holds eleven one-wave workgroups instead of twelve --
    // measured as 3 072 equal streams taking two rounds instead of one, profiles/r03_free_list_heads.txt)
    uint32_t winStats[16];
#if defined(FS_SER_PROFILE)
    uint32_t serStats[8];        // design study: [0] symbol start -> first context ready, [1] first-context coding + coder hand-off, [2] tail of the loop, [3] serial symbols, [4] failed window attempts (clocks / 64)
#endif       // windowed hit path: attempts, windows, symbols covered, rounds, redone windows; [8..15] phase clocks / 64 (FS_WIN_PROFILE builds)
};

// the three words of a context record as fetched (per-lane values, fetch still in flight): issue early, finish at first use
struct CtxRaw { uint32_t a, b, d; };

struct Coder {
    fs_gptr hb;                  // heap base - 1  (so that index ix lives at hb + ix)
    FS_LDS Shared* sh;
    uint32_t pText, UnitsStart, LoUnit, HiUnit;
    uint32_t MaxContext, FoundState;
    uint32_t fsSym, fsFreq, fsSucc;   // register copy of *FoundState (memory is always written through)
    uint32_t NumMasked, PrevSuccess, EscCount;
    int32_t BSumm, OrderFall, RunLength;
    uint32_t low, range, rLow, rHigh, rScale;
    fs_gptr out; uint32_t outCap, outPos;
    uint32_t queued, qTail, qHeadSeen;   // queued != 0: every coding step goes to the coder wave through sh->qA/qM (this wave never reads low/range)
    uint32_t pfCtx; CtxRaw pf;     // record of the next symbol's first context, requested ahead of this symbol's stores (0 = none)
    uint32_t inAhead;              // windowed path: the input has been pulled into the cache up to here
    uint32_t ldsHeads;             // the sub-allocator's list heads live in LDS (else behind the heap)
    uint32_t newCtx;               // the context the walk stands in was made by the last symbol's CreateSuccessors: a binary context, no window starts there
};

#if defined(FS_HEAP_STATS)      // design study (tools/heap_locality.cpp, host emulation only): which heap addresses the walk touches
  #define HP(ix) (m.hb + fs_heap_stat(ix))
#else
  #define HP(ix) (m.hb + (ix))
#endif
// PPM_CONTEXT: NumStats@0 Flags@1 SummFreq@2 iStats@4 iSuffix@8 ; oneState = STATE at +2
#define C_NS(c) fs_ld8(HP(c))
#define C_NS_SET(c, v) fs_st8(HP(c), (v))
#define C_FLAGS(c) fs_ld8(HP(c) + 1)
#define C_FLAGS_SET(c, v) fs_st8(HP(c) + 1, (v))
#define C_SF(c) fs_ld16(HP(c) + 2)
#define C_SF_SET(c, v) fs_st16(HP(c) + 2, (v))
#define C_STATS(c) fs_ld32(HP(c) + 4)
#define C_STATS_SET(c, v) fs_st32(HP(c) + 4, (v))
#define C_SUFF(c) fs_ld32(HP(c) + 8)
#define C_SUFF_SET(c, v) fs_st32(HP(c) + 8, (v))
#define C_ONE(c) ((c) + 2u)
// STATE: Symbol@0 Freq@1 iSuccessor@2
#define S_SYM(s) fs_ld8(HP(s))
#define S_FREQ(s) fs_ld8(HP(s) + 1)
#define S_SYMFREQ(s) fs_ld16(HP(s))
#define S_FREQ_SET(s, v) fs_st8(HP(s) + 1, (v))
#define S_SUCC(s) fs_ld32h(HP(s) + 2)
#define S_SUCC_SET(s, v) fs_st32h(HP(s) + 2, (v))
// BLK_NODE / MEM_BLK: Stamp@0 NextIndx@4 NU@8
#define B_STAMP(b) fs_ld32(HP(b))
#define B_STAMP_SET(b, v) fs_st32(HP(b), (v))
#define B_NEXT(b) fs_ld32(HP(b) + 4)
#define B_NEXT_SET(b, v) fs_st32(HP(b) + 4, (v))
#define B_NU(b) fs_ld32(HP(b) + 8)
#define B_NU_SET(b, v) fs_st32(HP(b) + 8, (v))
#define BL(i) (1u + HEADS_OFF + 12u * (uint32_t)(i))

FS_DEV void state_store(Coder& m, uint32_t s, uint32_t symfreq, uint32_t succ)
{ fs_st48(HP(s), symfreq, succ & 0xFFFFu, succ >> 16); }
FS_DEV void state_swap(Coder& m, uint32_t a, uint32_t b)
{
    uint32_t a0 = S_SYMFREQ(a), a1 = S_SUCC(a), b0 = S_SYMFREQ(b), b1 = S_SUCC(b);
    state_store(m, a, b0, b1); state_store(m, b, a0, a1);
}
FS_DEV void state_cpy(Coder& m, uint32_t d, uint32_t s) { state_store(m, d, S_SYMFREQ(s), S_SUCC(s)); }

// One context header = one 12-byte fetch (three dwords in flight together), then kept in SGPRs.
// kept packed as fetched (three SGPRs per record instead of seven): word 0 = NumStats | Flags << 8 | SummFreq << 16, where
// the SummFreq half is oneState {Symbol, Freq} in a binary context
struct Ctx {
    uint32_t a, w1 /* iStats, or oneState.iSuccessor */, suff;
    FS_DEV_M uint32_t ns() const { return a & 0xFFu; }
    FS_DEV_M uint32_t flags() const { return (a >> 8) & 0xFFu; }
    FS_DEV_M uint32_t sf() const { return a >> 16; }
    FS_DEV_M uint32_t oneSym() const { return (a >> 16) & 0xFFu; }
    FS_DEV_M uint32_t oneFreq() const { return a >> 24; }
    FS_DEV_M void set_sf(uint32_t v) { a = (a & 0xFFFFu) | (v << 16); }
    FS_DEV_M void set_one_freq(uint32_t v) { a = (a & 0x00FFFFFFu) | (v << 24); }
};
FS_DEV CtxRaw ctx_issue(Coder& m, uint32_t c)
{
    FS_CNT(g_ld[0]);
    fs_cgptr32 q = (fs_cgptr32)HP(c);
    CtxRaw r;
    r.a = q[0]; r.b = q[1]; r.d = q[2];
    FS_EMU_MEET();
    return r;
}
FS_DEV Ctx ctx_finish(const CtxRaw& x)
{
    const uint32_t a = FS_UNI(x.a), b = FS_UNI(x.b), d = FS_UNI(x.d);
    Ctx r; r.a = a; r.w1 = b; r.suff = d;
    return r;
}
FS_DEV Ctx ctx_load(Coder& m, uint32_t c) { return ctx_finish(ctx_issue(m, c)); }
// One state = three 16-bit fetches in flight together.
struct St { uint32_t sym, freq, succ; };
FS_DEV St st_load(Coder& m, uint32_t s)
{
    FS_CNT(g_ld[1]);
    fs_cgptr16 q = (fs_cgptr16)HP(s);
    uint32_t a = q[0], b = q[1], c = q[2];
    FS_EMU_MEET();
    a = FS_UNI(a); b = FS_UNI(b); c = FS_UNI(c);
    St r; r.sym = a & 0xFFu; r.freq = a >> 8; r.succ = b | (c << 16);
    return r;
}
FS_DEV void fs_reload(Coder& m) { if (m.FoundState) { const St t = st_load(m, m.FoundState); m.fsSym = t.sym; m.fsFreq = t.freq; m.fsSucc = t.succ; } }

// ---------------- sub-allocator ----------------
// (list number i: BLK_NODE::remove / insert / avail of SubAlloc.hpp:41-55.  The head -- first block of the list, 0 = empty -- lives
// in LDS in the form that walks single long streams (m.ldsHeads: two waves), behind the heap in HBM in the one-wave
// form: measured, profiles/r03_free_list_heads.txt -- a lone 7 M-symbol stream 976 -> 941 ms with the heads in LDS.  The one-wave
// form was first thought to lose by them (3 072 streams side by side 6.35 -> 4.43 G symbols/s); that loss was the workgroup's LDS
// passing 12 800 bytes, see Shared, and is gone at 12 796 bytes: 6.60 G symbols/s.)
FS_DEV uint32_t blk_head(Coder& m, uint32_t i) { return m.ldsHeads ? FS_LDS_RD(m.sh->blHead[i]) : B_NEXT(BL(i)); }
FS_DEV void blk_head_set(Coder& m, uint32_t i, uint32_t v) { if (m.ldsHeads) { m.sh->blHead[i] = v; FS_EMU_MEET(); } else B_NEXT_SET(BL(i), v); }
FS_DEV uint32_t blk_remove(Coder& m, uint32_t i)
{ const uint32_t p = blk_head(m, i); const uint32_t nx = B_NEXT(p); blk_head_set(m, i, nx); return p; }
FS_DEV void blk_insert(Coder& m, uint32_t i, uint32_t pv, uint32_t nu)
{ B_NEXT_SET(pv, blk_head(m, i)); blk_head_set(m, i, pv); B_STAMP_SET(pv, 0xFFFFFFFFu); B_NU_SET(pv, nu); }
FS_DEV bool blk_avail(Coder& m, uint32_t i) { return blk_head(m, i) != 0u; }

FS_DEV void SplitBlock(Coder& m, uint32_t pv, uint32_t oldI, uint32_t newI)
{
    uint32_t i, k, UDiff = (uint32_t)kIndx2Units[oldI] - kIndx2Units[newI];
    uint32_t p = pv + 12u * kIndx2Units[newI];
    if (kIndx2Units[i = kUnits2Indx[UDiff - 1]] != UDiff) {
        k = kIndx2Units[--i]; blk_insert(m, (i), p, k);
        p += 12u * k; UDiff -= k;
    }
    blk_insert(m, (kUnits2Indx[UDiff - 1]), p, UDiff);
}

FS_DEV void InitSubAllocator(Coder& m)
{
    // memset(BList, 0)
    for (uint32_t i = (uint32_t)FS_LANE(); i < (uint32_t)N_INDEXES + 2u; i += FS_WAVE) { m.sh->blHead[i] = 0u; *(fs_gptr32)(HP(BL(i)) + 4u) = 0u; }
    FS_WAVE_SYNC();
    m.pText = 1u; m.HiUnit = 1u + SA_SIZE;
    const uint32_t Diff = 12u * (SA_SIZE / 8 / UNIT_SIZE * 7);
    m.LoUnit = m.UnitsStart = m.HiUnit - Diff; m.sh->GlueCount = m.sh->GlueCount1 = 0;
}

FS_DEV_NOINLINE void GlueFreeBlocks(Coder& m)
{
    FS_REGION(7);
    uint32_t i, k, sz, p, p0, p1;
    const uint32_t s0 = N_INDEXES + 1;                 // the scratch list
    if (m.LoUnit != m.HiUnit) fs_st8(HP(m.LoUnit), 0);
    p0 = 0; blk_head_set(m, s0, 0u);                   // p0 = 0: the next block goes in right behind the head
    for (i = 0; i <= N_INDEXES; i++)
        while (blk_avail(m, (i))) {
            p = blk_remove(m, (i));
            if (!B_NU(p)) continue;
            while (B_STAMP(p1 = p + 12u * B_NU(p)) == 0xFFFFFFFFu) { B_NU_SET(p, B_NU(p) + B_NU(p1)); B_NU_SET(p1, 0); }
            if (p0 == 0u) { B_NEXT_SET(p, blk_head(m, s0)); blk_head_set(m, s0, p); }
            else { B_NEXT_SET(p, B_NEXT(p0)); B_NEXT_SET(p0, p); }
            p0 = p;
        }
    while (blk_avail(m, s0)) {
        p = blk_remove(m, s0); sz = B_NU(p);
        if (!sz) continue;
        for (; sz > 128; sz -= 128, p += 12u * 128) blk_insert(m, (N_INDEXES - 1), p, 128);
        if (kIndx2Units[i = kUnits2Indx[sz - 1]] != sz) { k = sz - kIndx2Units[--i]; blk_insert(m, (k - 1), p + 12u * (sz - k), k); }
        blk_insert(m, (i), p, kIndx2Units[i]);
    }
    { const uint32_t g1 = FS_LDS_RD(m.sh->GlueCount1); m.sh->GlueCount = 1u << (13 + g1); m.sh->GlueCount1 = g1 + 1u; }
}

FS_DEV_NOINLINE uint32_t AllocUnitsRare(Coder& m, uint32_t indx)
{
    FS_REGION(7);
    uint32_t i = indx;
    do {
        if (++i == N_INDEXES) {
            const uint32_t gc = FS_LDS_RD(m.sh->GlueCount); m.sh->GlueCount = gc - 1u;
            if (!gc) {
                GlueFreeBlocks(m);
                if (blk_avail(m, (i = indx))) return blk_remove(m, (i));
            } else {
                i = 12u * kIndx2Units[indx];
                return (m.UnitsStart - m.pText > i) ? (m.UnitsStart -= i) : 0u;
            }
        }
    } while (!blk_avail(m, (i)));
    uint32_t r = blk_remove(m, (i)); SplitBlock(m, r, i, indx);
    return r;
}

FS_DEV uint32_t AllocUnits(Coder& m, uint32_t NU)
{
    FS_REGION(6);
    uint32_t indx = kUnits2Indx[NU - 1];
    if (blk_avail(m, (indx))) return blk_remove(m, (indx));
    uint32_t r = m.LoUnit; m.LoUnit += 12u * kIndx2Units[indx];
    if (m.LoUnit <= m.HiUnit) return r;
    m.LoUnit -= 12u * kIndx2Units[indx]; return AllocUnitsRare(m, indx);
}

FS_DEV uint32_t AllocContext(Coder& m)
{
    FS_REGION(6);
    if (m.HiUnit != m.LoUnit) return (m.HiUnit -= UNIT_SIZE);
    return blk_avail(m, (0)) ? blk_remove(m, (0)) : AllocUnitsRare(m, 0);
}

FS_DEV uint32_t ExpandUnits(Coder& m, uint32_t oldPtr, uint32_t oldNU)
{
    FS_REGION(6);
    uint32_t i0 = kUnits2Indx[oldNU - 1], i1 = kUnits2Indx[oldNU - 1 + 1];
    if (i0 == i1) return oldPtr;
    uint32_t ptr = AllocUnits(m, oldNU + 1);
    if (ptr) { fs_wave_copy4(HP(ptr), HP(oldPtr), 12u * oldNU); blk_insert(m, (i0), oldPtr, oldNU); }
    return ptr;
}

FS_DEV uint32_t ShrinkUnits(Coder& m, uint32_t oldPtr, uint32_t oldNU, uint32_t newNU)
{
    FS_REGION(6);
    uint32_t i0 = kUnits2Indx[oldNU - 1], i1 = kUnits2Indx[newNU - 1];
    if (i0 == i1) return oldPtr;
    if (blk_avail(m, (i1))) {
        uint32_t ptr = blk_remove(m, (i1)); fs_wave_copy4(HP(ptr), HP(oldPtr), 12u * newNU);
        blk_insert(m, (i0), oldPtr, kIndx2Units[i0]);
        return ptr;
    }
    SplitBlock(m, oldPtr, i0, i1); return oldPtr;
}

FS_DEV void FreeUnits(Coder& m, uint32_t ptr, uint32_t NU)
{ uint32_t indx = kUnits2Indx[NU - 1]; blk_insert(m, (indx), ptr, kIndx2Units[indx]); }

// ---------------- range coder ----------------
// The carry-less coder of Coder.hpp:7-28.  One-wave form: the walking wave codes as it goes.  Two-wave form
// (m.queued): the model never needs low/range back -- an encoder is open loop -- so every coding step is written as one
// entry into a ring in LDS and a second wavefront of the workgroup, the coder wave, works them off in order while the
// model wave walks on (coder_wave below).  Entries, two words (A, M):
//   A < 2^31           a plain hit of the windowed path: A = cumulative frequency | frequency << 16 (7 bits) |
//                      PrevSuccess << 23 (model side only), M = the total; the coder wave turns the totals of a whole
//                      batch into reciprocals at once, one per lane
//   A = 2^31 | low     a step of the serial path: M = frequency | total << 16; a binary context's step is the same with
//                      the total 2^14 (Model.cpp:415-432: range >>= TOT_BITS is the division by it)
//   A = 0xFFFFFFFF     command M: stream start (header bytes), stream end (flush, size), exit
enum : uint32_t { CQ_SIZE = 256u, CQ_SERIAL = 0x80000000u, CQ_CMD = 0xFFFFFFFFu, CQ_START = 1u, CQ_END = 2u, CQ_EXIT = 3u,
                  // a range-coded stream's triples through the same ring (coder_wave<true>: the kernel with the windowed range coders).  Entry:
                  // A = cumulative frequency | (frequency & 0x3FFF) << 16 | CQ_RC ; M = total | (frequency >> 14) << 16
                  CQ_START_RC = 4u, CQ_END_RC = 5u, CQ_END_RC_BAD = 6u, CQ_RC = 1u << 30,
                  // a QVZ stream's symbols through the same ring (coder_wave<true, fsqvz::WaveCoder>, qvz_core.h).  Entry:
                  // A = counts below the symbol (20 bits) | (its count & 0x3FF) << 20 ; M = the context's total (20 bits) | (count >> 10) << 20 | CQ_QVZ
                  CQ_START_QVZ = 7u, CQ_END_QVZ = 8u, CQ_END_QVZ_BAD = 9u, CQ_QVZ = 1u << 31 };
// what coder_wave does with a QVZ stream's entries: nothing, in the kernels that carry no QVZ coder on their coder wave
struct NoQvz {
    static constexpr bool on = false;
    struct State {};
    FS_DEV_M static void start(State&, fs_gptr, uint32_t) {}
    FS_DEV_M static void prepare(uint32_t, uint32_t, uint32_t&, uint32_t&, uint32_t&, uint32_t&) {}
    FS_DEV_M static void code(State&, uint32_t, uint32_t, uint32_t, uint32_t) {}
    FS_DEV_M static uint32_t finish(State&) { return 0u; }
};

FS_DEV void put_byte(Coder& m, uint32_t c) { if (m.outPos < m.outCap) fs_st8(m.out + m.outPos, c); m.outPos += (m.outPos < m.outCap); }
// (a range-coded stream's byte: its size counts on past the capacity, as rc_core.h's put does)
FS_DEV void rc_put(Coder& m, uint32_t c) { if (m.outPos < m.outCap) fs_st8(m.out + m.outPos, c); m.outPos++; }
FS_DEV void rc_shift_out(Coder& m)
{
    while ((m.low ^ (m.low + m.range)) < TOP || (m.range < BOT && ((m.range = (0u - m.low) & (BOT - 1)), true))) {
        put_byte(m, m.low >> 24);
        m.range <<= 8; m.low <<= 8;
    }
}

#if FS_WIDE
// polls of a word the other wave of the workgroup writes
#if defined(__HIP_DEVICE_COMPILE__)
  #define FS_Q_LOAD(w) FS_UNI(__hip_atomic_load(&(w), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP))
  #define FS_Q_STORE(w, v) do { if (FS_LANE() == 0) __hip_atomic_store(&(w), (uint32_t)(v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)
  #define FS_Q_IDLE() __builtin_amdgcn_s_sleep(8)
#else
  #define FS_Q_LOAD(w) fs_q_load(&(w))
  FS_DEV uint32_t fs_q_load(const uint32_t* p) { FS_EMU_MEET(); const uint32_t v = *(const volatile uint32_t*)p; FS_EMU_MEET(); return v; }
  #define FS_Q_STORE(w, v) do { FS_EMU_MEET(); if (FS_LANE() == 0) *(volatile uint32_t*)&(w) = (uint32_t)(v); FS_EMU_MEET(); } while (0)
  #define FS_Q_IDLE() ((void)0)
#endif
// room for `need` more entries (the consumer frees a batch as soon as it has it in registers)
FS_DEV void cq_room(Coder& m, uint32_t need)
{
    while (m.qTail + need - m.qHeadSeen > CQ_SIZE) { m.qHeadSeen = FS_Q_LOAD(m.sh->qHead); if (m.qTail + need - m.qHeadSeen > CQ_SIZE) FS_Q_IDLE(); }
}
FS_DEV void cq_push(Coder& m, uint32_t A, uint32_t M)
{
    cq_room(m, 1u);
    if (FS_LANE() == 0) { m.sh->qA[m.qTail & (CQ_SIZE - 1u)] = A; m.sh->qM[m.qTail & (CQ_SIZE - 1u)] = M; }
    m.qTail += 1u;
    FS_Q_STORE(m.sh->qTail, m.qTail);
}
// L entries, one per lane
FS_DEV void cq_push_lanes(Coder& m, uint32_t A, uint32_t M, uint32_t L)
{
    cq_room(m, L);
    if ((uint32_t)FS_LANE() < L) { const uint32_t at = (m.qTail + (uint32_t)FS_LANE()) & (CQ_SIZE - 1u); m.sh->qA[at] = A; m.sh->qM[at] = M; }
    m.qTail += L;
    FS_Q_STORE(m.sh->qTail, m.qTail);
}
// the coder wave has opened every stream up to number `upTo` - 1 (their mailboxes are read)
FS_DEV void cq_wait_starts(Coder& m, uint32_t upTo) { while ((int32_t)(FS_Q_LOAD(m.sh->qStarts) - upTo) < 0) FS_Q_IDLE(); }
// the coder wave takes over the issue priority of the stream whose pass it runs (the launch ends with its longest stream)
FS_DEV void cq_take_priority(uint32_t p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (p >= 3u) __builtin_amdgcn_s_setprio(3); else if (p == 2u) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#else
    (void)p;
#endif
}
// the model wave's last word: the coder wave returns
FS_DEV void cq_send_exit(FS_LDS Shared* sh, uint32_t qTail)
{ Coder m; m.sh = sh; m.qTail = qTail; m.qHeadSeen = qTail - CQ_SIZE; cq_push(m, CQ_CMD, CQ_EXIT); }
#else
FS_DEV void cq_push(Coder&, uint32_t, uint32_t) {}
FS_DEV void cq_send_exit(FS_LDS Shared*, uint32_t) {}
#if defined(__HIPCC__)      // (the compiler's host pass over a kernel that names the template: it must find a callable declaration)
template <bool RC, class QV = NoQvz> __host__ __device__ inline void coder_wave(FS_LDS Shared*) {}
#else
template <bool RC, class QV = NoQvz> FS_DEV void coder_wave(FS_LDS Shared*) {}
#endif
#endif

// after a coding step (Model.cpp:569, 580)
FS_DEV void rc_normalize(Coder& m) { if (!m.queued) rc_shift_out(m); }
// one step with (rLow, rHigh, rScale)
FS_DEV void rc_encode(Coder& m)
{
    if (m.queued) { cq_push(m, CQ_SERIAL | m.rLow, (m.rHigh - m.rLow) | (m.rScale << 16)); return; }
    m.low += m.rLow * (m.range /= m.rScale); m.range *= m.rHigh - m.rLow;
}
// a binary context's step (rcBinStart / rcBinCorrect0 / rcBinCorrect1): the symbol has probability bs / 2^14
FS_DEV void rc_encode_bin(Coder& m, uint32_t bs, bool hit)
{
    if (m.queued) { cq_push(m, CQ_SERIAL | (hit ? 0u : bs), (hit ? bs : (uint32_t)BIN_SCALE - bs) | ((uint32_t)BIN_SCALE << 16)); return; }
    const uint32_t tmp = bs * (m.range >>= TOT_BITS);
    if (hit) m.range = tmp; else { m.low += tmp; m.range *= (uint32_t)BIN_SCALE - bs; }
}

// ---------------- model ----------------
FS_DEV void clear_mask(Coder& m)
{
    for (uint32_t i = (uint32_t)FS_LANE(); i < 64u; i += FS_WAVE) ((FS_LDS uint32_t*)m.sh->CharMask)[i] = 0u;
    FS_WAVE_SYNC();
    m.EscCount = 1;
}

FS_DEV void StartModelRare(Coder& m)
{
    FS_REGION(9);
    clear_mask(m);
    m.OrderFall = MAX_ORDER;
    InitSubAllocator(m);
    m.RunLength = INIT_RL;
    m.MaxContext = AllocContext(m);
    C_NS_SET(m.MaxContext, 255); C_SF_SET(m.MaxContext, 257);
    const uint32_t st = AllocUnits(m, 128);
    C_STATS_SET(m.MaxContext, st);
    m.PrevSuccess = 0; C_SUFF_SET(m.MaxContext, 0); C_FLAGS_SET(m.MaxContext, 0);
    for (uint32_t i = (uint32_t)FS_LANE(); i < 256u; i += FS_WAVE) {       // {Symbol=i, Freq=1, iSuccessor=0}
        fs_gptr16 q = (fs_gptr16)(HP(st) + 6u * i);
        q[0] = (uint16_t)(i | 0x100u); q[1] = 0; q[2] = 0;
    }
    // binary SEE contexts: BinSumm[i][k] = BIN_SCALE - 128*clamp(sum coef)/i2f[i],  i2f[i] = (#k: QTable[k] <= i) + 1
    for (uint32_t e = (uint32_t)FS_LANE(); e < 25u * 64u; e += FS_WAVE) {
        const uint32_t i = e >> 6, k = e & 63u;
        int s = 0;
        for (int b = 0; b < 6; ++b) s += kEscCoef[2 * b + ((k >> b) & 1u)];
        s = 128 * (s < 32 ? 32 : (s > 256 - 32 ? 256 - 32 : s));
        uint32_t kk = 0; while (FS_UNI(m.sh->QT[kk]) <= i) kk++;              // first k with QTable[k] > i
        m.sh->BinSumm[e] = (uint16_t)(BIN_SCALE - s / (int)(kk + 1));
    }
    for (uint32_t e = (uint32_t)FS_LANE(); e < 23u * 32u; e += FS_WAVE) {
        const uint32_t i = e >> 5;                                     // init(8*i+5): Summ = v << 3, Shift = 3, Count = 7
        m.sh->SEE2[e] = ((8u * i + 5u) << (PERIOD_BITS - 4)) | ((uint32_t)(PERIOD_BITS - 4) << 16) | (7u << 24);
    }
    FS_WAVE_SYNC();
}

FS_DEV void RestoreModelRare(Coder& m) { m.pText = 1u; StartModelRare(m); m.EscCount = 0; m.sh->restarts = FS_LDS_RD(m.sh->restarts) + 1u; }

// per-lane view of 64 consecutive states of a context: one fetch for the whole list
struct LaneStates { uint32_t sf, succ; bool valid; };
FS_DEV LaneStates lane_states(Coder& m, uint32_t stats, uint32_t ns, uint32_t base)
{
#if defined(FS_COUNTERS) && !defined(__HIP_DEVICE_COMPILE__)
    if ((base & 63u) == 0) { FS_CNT(g_ld[2]); FS_CNT(g_ld[2]); FS_CNT(g_ld[2]); }      // what the 64-lane build issues: three fetches per 64 states
#endif
    LaneStates r; const uint32_t i = base + (uint32_t)FS_LANE();
    r.valid = i <= ns;
    // lanes past the end re-read the last state: an unconditional fetch keeps this off the exec-masked path
    const uint32_t ii = r.valid ? i : ns;
    fs_cgptr16 q = (fs_cgptr16)HP(stats + 6u * ii); const uint32_t a = q[0], b = q[1], c = q[2];
    FS_EMU_MEET();
    r.sf = r.valid ? a : 0u; r.succ = r.valid ? (b | (c << 16)) : 0u;
    return r;
}

// the state of `sym` in a multi-symbol context (it must exist), with its record and its predecessor's
struct Hit { uint32_t p, freq, succ, prevSf, prevSucc; };
FS_DEV Hit find_in(Coder& m, const Ctx& pc, uint32_t sym)
{
    Hit h; h.p = pc.w1; h.freq = 0; h.succ = 0; h.prevSf = 0; h.prevSucc = 0;
#if FS_WIDE
    if (FS_UNI(pc.ns()) < FS_WAVE) {                          // the whole list in one fetch: no loop
        const LaneStates ls = lane_states(m, pc.w1, pc.ns(), 0);
        const uint64_t hit = fs_ballot(ls.valid && (ls.sf & 0xFFu) == sym);
        const uint32_t k = hit ? fs_ctz64(hit) : 0u;
        h.p = pc.w1 + 6u * k;
        h.freq = FS_UNI(fs_readlane(ls.sf >> 8, k)); h.succ = FS_UNI(fs_readlane(ls.succ, k));
        const uint32_t kp = k ? k - 1u : 0u;
        h.prevSf = k ? FS_UNI(fs_readlane(ls.sf, kp)) : 0u; h.prevSucc = k ? FS_UNI(fs_readlane(ls.succ, kp)) : 0u;
        return h;
    }
#endif
    for (uint32_t base = 0;; base += FS_WAVE) {
        const LaneStates ls = lane_states(m, pc.w1, pc.ns(), base);
        const uint64_t hit = fs_ballot(ls.valid && (ls.sf & 0xFFu) == sym);
        if (hit) {
            const uint32_t k = fs_ctz64(hit);
            h.p = pc.w1 + 6u * (base + k);
            h.freq = FS_UNI(fs_readlane(ls.sf >> 8, k)); h.succ = FS_UNI(fs_readlane(ls.succ, k));
            if (k > 0) { h.prevSf = FS_UNI(fs_readlane(ls.sf, k - 1)); h.prevSucc = FS_UNI(fs_readlane(ls.succ, k - 1)); }
            else if (base > 0) { const St t = st_load(m, h.p - 6); h.prevSf = t.sym | (t.freq << 8); h.prevSucc = t.succ; }
            return h;
        }
        if (base + FS_WAVE > pc.ns()) return h;            // unreachable for well-formed models; never spin
    }
}

// CreateSuccessors (Model.cpp:282-337).  p/pSucc: state to start from in the suffix of pc (0 = none) and its successor.
// haveRecs: pcRec / sufRec are register copies of the records of pc and (when p != 0) of its suffix
FS_DEV uint32_t CreateSuccessors(Coder& m, bool Skip, uint32_t p, uint32_t pSucc, uint32_t pc, uint32_t fsSym, uint32_t fsSucc,
                                 bool haveRecs, const Ctx& pcRec, const Ctx& sufRec)
{
    FS_REGION(3);
    FS_PATH(g_path[9]);
    const uint32_t iUpBranch = fsSucc;
    uint32_t ps[MAX_ORDER + 1]; uint32_t pps = 0;
    uint32_t cf, s0, tmp;
    uint32_t sym = fsSym;
    Ctx P; if (haveRecs) P = pcRec; else P = ctx_load(m, pc);
    bool toLoop = true;
    if (!Skip) {
        ps[pps++] = m.FoundState;
        if (!P.suff) toLoop = false;
    }
    if (toLoop) {
        bool first = (p != 0);
        if (first) { pc = P.suff; if (haveRecs) P = sufRec; else P = ctx_load(m, pc); }
        do {
            if (!first) {
                pc = P.suff; P = ctx_load(m, pc);
                if (P.ns()) {
                    const Hit h = find_in(m, P, sym);
                    p = h.p; pSucc = h.succ;
                    tmp = (h.freq < MAX_FREQ);
                    S_FREQ_SET(p, h.freq + tmp); P.set_sf(P.sf() + tmp); C_SF_SET(pc, P.sf());
                } else {
                    p = C_ONE(pc); pSucc = P.w1;
                    const uint32_t sufNs = fs_ld8(HP(P.suff));
                    S_FREQ_SET(p, P.oneFreq() + ((!sufNs) & (P.oneFreq() < 11)));
                }
            }
            first = false;
            if (pSucc != iUpBranch) { pc = pSucc; P = ctx_load(m, pc); break; }
            ps[pps++] = p;
        } while (P.suff);
    }
    if (pps == 0) return pc;
    uint32_t ctFlags = 0x10u * (sym >= 0x40);
    sym = fs_ld8(HP(iUpBranch));
    ctFlags |= 0x08u * (sym >= 0x40);
    uint32_t ctFreq;
    if (P.ns()) {
        const Hit h = find_in(m, P, sym);
        s0 = P.sf() - P.ns() - (cf = h.freq - 1u);
        cf = 1 + ((2 * cf <= s0) ? (uint32_t)(12 * cf > s0) : ((cf + 2 * s0) / s0));
        ctFreq = (cf < 7) ? cf : 7;
    } else ctFreq = P.oneFreq();
    const uint32_t w0 = (ctFlags << 8) | (sym << 16) | (ctFreq << 24);   // NumStats=0, Flags, oneState{Symbol,Freq}
    do {
        const uint32_t pc1 = AllocContext(m);
        if (!pc1) return 0;
        fs_st32(HP(pc1), w0); fs_st32(HP(pc1) + 4, iUpBranch + 1u); fs_st32(HP(pc1) + 8, pc);
        S_SUCC_SET(ps[--pps], pc = pc1);
    } while (pps != 0);
    m.newCtx = pc;
    return pc;
}

// ReduceOrder (Model.cpp:209-243)
FS_DEV uint32_t ReduceOrder(Coder& m, uint32_t p, uint32_t pSucc, uint32_t pc)
{
    FS_REGION(4);
    FS_PATH(g_path[10]);
    uint32_t tmp; const uint32_t pc1 = pc;
    const uint32_t iUpBranch = m.pText; S_SUCC_SET(m.FoundState, iUpBranch);
    const uint32_t sym = m.fsSym; m.OrderFall++;
    Ctx P = ctx_load(m, pc);
    bool first = (p != 0);
    if (first) { pc = P.suff; P = ctx_load(m, pc); }
    for (;;) {
        if (!first) {
            if (!P.suff) return pc;
            pc = P.suff; P = ctx_load(m, pc);
            if (P.ns()) {
                const Hit h = find_in(m, P, sym);
                p = h.p; pSucc = h.succ;
                tmp = 2u * (h.freq < MAX_FREQ - 3);
                S_FREQ_SET(p, h.freq + tmp); P.set_sf(P.sf() + tmp); C_SF_SET(pc, P.sf());
            } else { p = C_ONE(pc); pSucc = P.w1; S_FREQ_SET(p, P.oneFreq() + (P.oneFreq() < 11)); }
        }
        first = false;
        if (pSucc) break;
        S_SUCC_SET(p, iUpBranch); m.OrderFall++;
    }
    if (pSucc <= iUpBranch) {
        const uint32_t p1 = m.FoundState; m.FoundState = p;
        pSucc = CreateSuccessors(m, false, 0, 0, pc, sym, pSucc, true, P, P);
        S_SUCC_SET(p, pSucc);
        m.FoundState = p1;
    }
    if (m.OrderFall == 1 && pc1 == m.MaxContext) { S_SUCC_SET(m.FoundState, pSucc); m.pText--; }
    return pSucc;
}

// The reference's rescale (Model.cpp:246-280) is a move-to-front of the found state, a halving pass and an insertion
// sort, one state at a time.  Its result is a stable sort of the rotated list by the halved frequencies (the found
// state stays first: it is the only one above MAX_FREQ), so for contexts of up to 64 states every lane takes one state,
// computes its final position by counting, and the list goes back in one step; the sums are wave reductions.
FS_DEV_NOINLINE void rescale_serial(Coder& m, uint32_t c);
FS_DEV_NOINLINE void rescale(Coder& m, uint32_t c)
{
    const Ctx R = ctx_load(m, c);
    const uint32_t ns = R.ns(), stats = R.w1;
    if (ns >= 64u) { rescale_serial(m, c); return; }
    FS_REGION(5);
    FS_PATH(g_path[11]);
    const uint32_t kf = (m.FoundState - stats) / 6u, a0 = (m.OrderFall != 0);
    uint32_t sumF, summ, zeros, f0, nf0, sym0, succ0; bool hiAny;
#if FS_WIDE
    {
        const uint32_t i = (uint32_t)FS_LANE();
        const LaneStates ls = lane_states(m, stats, ns, 0);
        const uint32_t sym = ls.sf & 0xFFu, f = ls.sf >> 8;
        const uint32_t nf = ls.valid ? (f + a0) >> 1 : 0u;
        sumF = fs_wave_sum8(f, ls.valid); summ = fs_wave_sum8(nf, ls.valid);
        hiAny = fs_ballot(ls.valid && i != kf && nf != 0u && sym >= 0x40u) != 0ull;
        zeros = fs_popc64(fs_ballot(ls.valid && i != kf && nf == 0u));
        const uint32_t r = i == kf ? 0u : (i < kf ? i + 1u : i);             // place after the move-to-front
        uint32_t pos = 1u;
        for (uint32_t j = 0; j <= ns; ++j) {                                  // uniform loop, one comparison per lane
            const uint32_t nj = fs_readlane(nf, j), rj = j == kf ? 0u : (j < kf ? j + 1u : j);
            pos += (j != kf && (nj > nf || (nj == nf && rj < r))) ? 1u : 0u;
        }
        if (i == kf) pos = 0u;
        f0 = fs_readlane(f, kf); nf0 = fs_readlane(nf, kf); sym0 = fs_readlane(sym, kf); succ0 = fs_readlane(ls.succ, kf);
        if (ls.valid) {
            fs_gptr16 q = (fs_gptr16)HP(stats + 6u * pos);
            q[0] = (uint16_t)(sym | (nf << 8)); q[1] = (uint16_t)ls.succ; q[2] = (uint16_t)(ls.succ >> 16);
        }
        FS_WAVE_SYNC();
    }
#else
    {
        uint32_t sym[64], f[64], nf[64], succ[64], r[64];
        for (uint32_t i = 0; i <= ns; ++i) { const St t = st_load(m, stats + 6u * i); sym[i] = t.sym; f[i] = t.freq; succ[i] = t.succ; nf[i] = (t.freq + a0) >> 1; r[i] = i == kf ? 0u : (i < kf ? i + 1u : i); }
        sumF = summ = zeros = 0; hiAny = false;
        for (uint32_t i = 0; i <= ns; ++i) { sumF += f[i]; summ += nf[i]; if (i != kf) { hiAny |= nf[i] != 0u && sym[i] >= 0x40u; zeros += nf[i] == 0u; } }
        for (uint32_t i = 0; i <= ns; ++i) {
            uint32_t pos = 1u;
            for (uint32_t j = 0; j <= ns; ++j) pos += (j != kf && (nf[j] > nf[i] || (nf[j] == nf[i] && r[j] < r[i]))) ? 1u : 0u;
            if (i == kf) pos = 0u;
            state_store(m, stats + 6u * pos, sym[i] | (nf[i] << 8), succ[i]);
        }
        f0 = f[kf]; nf0 = nf[kf]; sym0 = sym[kf]; succ0 = succ[kf];
    }
#endif
    uint32_t flags = (R.flags() & 0x14u) | (hiAny ? 0x08u : 0u);
    uint32_t EscFreq = R.sf() - sumF, nsNew = ns, statsNew = stats, a;
    if (zeros) {
        EscFreq += zeros;
        const uint32_t oldNU = (ns + 2u) >> 1;
        nsNew = ns - zeros;
        if (nsNew == 0) {                                                      // one state left: back to a binary context
            uint32_t tf = (2u * nf0 + EscFreq - 1) / EscFreq;
            if (tf > MAX_FREQ / 3) tf = MAX_FREQ / 3;
            FreeUnits(m, stats, oldNU);
            fs_st32(HP(c), ((flags & 0x18u) << 8) | (sym0 << 16) | (tf << 24));
            fs_st32(HP(c) + 4, succ0);
            m.FoundState = C_ONE(c); return;
        }
        statsNew = ShrinkUnits(m, stats, oldNU, (nsNew + 2u) >> 1);
    }
    summ += (EscFreq + 1) >> 1;
    if (m.OrderFall || (flags & 0x04u) == 0) {
        const uint32_t sf = R.sf() - EscFreq;
        a = sf - f0;
        a = (f0 * summ - sf * nf0 + a - 1) / a;
        a = a < 2u ? 2u : (a > MAX_FREQ / 2u - 18u ? MAX_FREQ / 2u - 18u : a);
    } else a = 2;
    m.FoundState = statsNew;
    S_FREQ_SET(statsNew, nf0 + a);
    fs_st32(HP(c), (nsNew & 0xFFu) | ((flags | 0x04u) << 8) | (((summ + a) & 0xFFFFu) << 16));
    if (statsNew != stats) fs_st32(HP(c) + 4, statsNew);
}

FS_DEV_NOINLINE void rescale_serial(Coder& m, uint32_t c)
{
    FS_REGION(5);
    FS_PATH(g_path[11]);
    uint32_t f0, sf, EscFreq, a = (m.OrderFall != 0), i = C_NS(c);
    uint32_t p1, p;
    C_FLAGS_SET(c, C_FLAGS(c) & 0x14u);
    const uint32_t stats = C_STATS(c);
    for (p = m.FoundState; p != stats; p -= 6) state_swap(m, p, p - 6);
    f0 = S_FREQ(p); sf = C_SF(c);
    EscFreq = sf - f0;
    uint32_t nf = (f0 + a) >> 1; S_FREQ_SET(p, nf);
    uint32_t summ = nf, flags = C_FLAGS(c);
    do {
        p += 6; const uint32_t fr = S_FREQ(p); EscFreq -= fr;
        nf = (fr + a) >> 1; S_FREQ_SET(p, nf); summ += nf;
        if (nf) flags |= 0x08u * (S_SYM(p) >= 0x40);
        if (nf > S_FREQ(p - 6)) {
            const uint32_t t0 = S_SYMFREQ(p), t1 = S_SUCC(p);
            p1 = p;
            do { state_cpy(m, p1, p1 - 6); p1 -= 6; } while (nf > S_FREQ(p1 - 6));
            state_store(m, p1, t0, t1);
        }
    } while (--i);
    C_FLAGS_SET(c, flags);
    if (S_FREQ(p) == 0) {
        do { i++; p -= 6; } while (S_FREQ(p) == 0);
        EscFreq += i; a = (C_NS(c) + 2u) >> 1;
        const uint32_t ns = C_NS(c) - i; C_NS_SET(c, ns);
        if (ns == 0) {
            const uint32_t t0 = S_SYMFREQ(stats), t1 = S_SUCC(stats);
            C_FLAGS_SET(c, C_FLAGS(c) & 0x18u);
            uint32_t tf = (2u * (t0 >> 8) + EscFreq - 1) / EscFreq;
            if (tf > MAX_FREQ / 3) tf = MAX_FREQ / 3;
            FreeUnits(m, stats, a);
            state_store(m, C_ONE(c), (t0 & 0xFFu) | (tf << 8), t1);
            m.FoundState = C_ONE(c); return;
        }
        C_STATS_SET(c, ShrinkUnits(m, stats, a, (ns + 2u) >> 1));
    }
    summ += (EscFreq + 1) >> 1;
    if (m.OrderFall || (C_FLAGS(c) & 0x04u) == 0) {
        a = (sf -= EscFreq) - f0;
        a = (f0 * summ - sf * S_FREQ(C_STATS(c)) + a - 1) / a;
        a = a < 2u ? 2u : (a > MAX_FREQ / 2u - 18u ? MAX_FREQ / 2u - 18u : a);
    } else a = 2;
    m.FoundState = C_STATS(c);
    S_FREQ_SET(m.FoundState, S_FREQ(m.FoundState) + a); C_SF_SET(c, summ + a);
    C_FLAGS_SET(c, C_FLAGS(c) | 0x04u);
}

FS_DEV void UpdateModel(Coder& m, uint32_t MinContext, const Ctx& mcMin, bool haveSuf, const Ctx& sufRec)
{
    FS_REGION(2);
    FS_PATH(g_path[8]);
    const uint32_t FSymbol = m.fsSym, FFreq = m.fsFreq;
    uint32_t iSuccessor, iFSuccessor = m.fsSucc;
    uint32_t ns1, ns, cf, sf, s0, pc, p = 0, pSucc = 0;
    bool restart = false;
    // mcMin: the caller's register copy of the context, kept in step with the stores of the coding step
    Ctx P; P.a = 0; P.w1 = 0; P.suff = 0;
    if (mcMin.suff) {
        pc = mcMin.suff;
        if (haveSuf) P = sufRec; else P = ctx_load(m, pc);
        if (P.ns()) {
            Hit h = find_in(m, P, FSymbol);
            p = h.p; pSucc = h.succ;
            cf = h.freq < MAX_FREQ ? 1 + (FFreq < 4 * 8) : 0;
            if (p != P.w1 && h.freq >= (h.prevSf >> 8)) {      // bubble one position up, the frequency update folded in
                state_store(m, p - 6, FSymbol | ((h.freq + cf) << 8), h.succ);
                state_store(m, p, h.prevSf, h.prevSucc);
                p -= 6;
            } else if (cf) S_FREQ_SET(p, h.freq + cf);
            if (cf) { P.set_sf(P.sf() + cf); C_SF_SET(pc, P.sf()); }
        } else { p = C_ONE(pc); pSucc = P.w1; P.set_one_freq(P.oneFreq() + (P.oneFreq() < 11)); S_FREQ_SET(p, P.oneFreq()); }
    }
    pc = m.MaxContext;
    if (!m.OrderFall && iFSuccessor) {
        const uint32_t sx = CreateSuccessors(m, true, p, pSucc, MinContext, FSymbol, iFSuccessor, true, mcMin, P);
        S_SUCC_SET(m.FoundState, sx);
        if (!sx) { RestoreModelRare(m); return; }
        m.MaxContext = sx; return;
    }
    fs_st8(HP(m.pText), FSymbol); m.pText++; iSuccessor = m.pText;
    if (m.pText >= m.UnitsStart) { RestoreModelRare(m); return; }
    if (iFSuccessor) {
        if (iFSuccessor < m.UnitsStart) iFSuccessor = CreateSuccessors(m, false, p, pSucc, MinContext, FSymbol, iFSuccessor, true, mcMin, P);
    } else iFSuccessor = ReduceOrder(m, p, pSucc, MinContext);
    if (!iFSuccessor) { RestoreModelRare(m); return; }
    if (!--m.OrderFall) { iSuccessor = iFSuccessor; m.pText -= (m.MaxContext != MinContext); }
    s0 = mcMin.sf() - FFreq; ns = mcMin.ns();
    const uint32_t Flag = 0x08u * (FSymbol >= 0x40);
    // one record fetch per context on the way down to MinContext; NumStats, Flags and SummFreq go back as one word
    while (pc != MinContext) {
        uint32_t summ, stats;
        FS_PATH(g_path[12]);
        const Ctx R = ctx_load(m, pc);
        if ((ns1 = R.ns()) != 0) {
            stats = R.w1;
            if ((ns1 & 1) != 0) {
                p = ExpandUnits(m, stats, (ns1 + 1) >> 1);
                if (!p) { restart = true; break; }
                stats = p;
            }
            summ = R.sf() + (FS_UNI(m.sh->QT[ns + 4]) >> 3);
        } else {
            p = AllocUnits(m, 1);
            if (!p) { restart = true; break; }
            uint32_t fr = R.oneFreq();
            fr = (fr <= MAX_FREQ / 3) ? (2 * fr - 1) : (MAX_FREQ - 15);
            state_store(m, p, R.oneSym() | (fr << 8), R.w1);           // oneState moves into its own unit
            stats = p;
            summ = fr + (ns > 1) + kExpEscape[FS_UNI(m.sh->QT[m.BSumm >> 8])];
        }
        cf = 2 * FFreq * (summ + 4u); sf = s0 + summ;
        if (cf <= 6 * sf) { cf = 1 + (cf > sf) + (cf > 3 * sf); summ += 4; }
        else { cf = 4 + (cf > 8 * sf) + (cf > 10 * sf) + (cf > 13 * sf); summ += cf; }
        ns1 += 1;
        state_store(m, stats + 6u * ns1, FSymbol | (cf << 8), iSuccessor);
        fs_st32(HP(pc), (ns1 & 0xFFu) | (((R.flags() | Flag) & 0xFFu) << 8) | ((summ & 0xFFFFu) << 16));
        if (stats != R.w1 || R.ns() == 0) fs_st32(HP(pc) + 4, stats);
        pc = R.suff;
    }
    if (restart) { RestoreModelRare(m); return; }
    m.MaxContext = iFSuccessor;
}

// The next symbol starts in the found state's successor when the model needs no update (main loop: !OrderFall and a
// real context).  Its record is requested HERE, ahead of this symbol's stores: the memory pipeline completes in issue
// order, so a fetch issued behind the stores would also wait for their acknowledgements.  A context that succeeds
// itself needs no fetch at all: the caller's register copy is kept current.
FS_DEV void prefetch_successor(Coder& m, uint32_t c, uint32_t succ)
{
    m.pfCtx = 0;
    if (!m.OrderFall && succ >= m.UnitsStart && succ != c) { m.pf = ctx_issue(m, succ); m.pfCtx = succ; }
}

// sufRec/sufCtx (here and in encodeSymbol2): the suffix context's record comes with the NumStats these functions need
// from it, and is handed on -- the next escape step and UpdateModel start from it instead of fetching it again.
FS_DEV void encodeBinSymbol(Coder& m, uint32_t c, Ctx& mc, int symbol, Ctx& sufRec, uint32_t& sufCtx)
{
    FS_REGION(1);
    const uint32_t rs = C_ONE(c);
    sufRec = ctx_load(m, mc.suff); sufCtx = mc.suff;
    const uint32_t sufNs = sufRec.ns();
    const uint32_t idx = FS_UNI(m.sh->QT[mc.oneFreq() - 1]) * 64u + NS2BSIndx(sufNs) + m.PrevSuccess + mc.flags() + (uint32_t)((m.RunLength >> 26) & 0x20);
    uint32_t bs = FS_LDS_RD(m.sh->BinSumm[idx]);
    m.BSumm = (int32_t)bs;
    rc_encode_bin(m, bs, (int)mc.oneSym() == symbol);
    bs -= (bs + ROUND) >> PERIOD_BITS;
    if ((int)mc.oneSym() == symbol) {
        prefetch_successor(m, c, mc.w1);
        bs += INTERVAL;
        const uint32_t nf = mc.oneFreq() + (mc.oneFreq() < 196);
        m.FoundState = rs; S_FREQ_SET(rs, nf);
        m.fsSym = mc.oneSym(); m.fsFreq = nf; m.fsSucc = mc.w1;
        mc.set_one_freq(nf);          // the caller's copy of the record follows the store
        m.RunLength++; m.PrevSuccess = 1;
    } else {
        m.sh->CharMask[mc.oneSym()] = (uint8_t)m.EscCount;
        m.NumMasked = m.PrevSuccess = 0; m.FoundState = 0;
    }
    m.sh->BinSumm[idx] = (uint16_t)bs;
}

FS_DEV void encodeSymbol1(Coder& m, uint32_t c, Ctx& mc, int symbol)
{
    FS_REGION(1);
    const uint32_t stats = mc.w1, ns = mc.ns();
    m.rScale = mc.sf();
    uint32_t LoCnt = 0, p = 0, k = 0, base = 0; bool found = false;
    LaneStates ls = lane_states(m, stats, ns, 0);
#if FS_WIDE
    if (FS_UNI(ns) < FS_WAVE) {                              // the usual case: the whole list is in the lanes, no loop
        const uint64_t hit = fs_ballot(ls.valid && (int)(ls.sf & 0xFFu) == symbol);
        if (hit != 0) { k = FS_UNI(fs_ctz64(hit)); if (k != 0) LoCnt = fs_wave_sum8(ls.sf >> 8, ls.valid && (uint32_t)FS_LANE() < k); p = stats + 6u * k; found = true; }
        else LoCnt = fs_wave_sum8(ls.sf >> 8, ls.valid);
    } else
#endif
    for (;;) {
        const uint64_t hit = fs_ballot(ls.valid && (int)(ls.sf & 0xFFu) == symbol);
        if (hit) { k = fs_ctz64(hit); if (k) LoCnt += fs_wave_sum8(ls.sf >> 8, ls.valid && (uint32_t)FS_LANE() < k); p = stats + 6u * (base + k); found = true; break; }
        LoCnt += fs_wave_sum8(ls.sf >> 8, ls.valid);
        if (base + FS_WAVE > ns) break;
        base += FS_WAVE; ls = lane_states(m, stats, ns, base);
    }
    if (FS_UNI((uint32_t)found) == 0) {                     // escape: mask every symbol of the context
        m.PrevSuccess = 0; m.rLow = LoCnt;
        const uint8_t esc = (uint8_t)m.EscCount;
        if (ns < FS_WAVE) { if (ls.valid) m.sh->CharMask[ls.sf & 0xFFu] = esc; }
        else for (uint32_t b2 = 0; b2 <= ns; b2 += FS_WAVE) { const uint32_t i = b2 + (uint32_t)FS_LANE(); if (i <= ns) m.sh->CharMask[*HP(stats + 6u * i)] = esc; }
        FS_WAVE_SYNC();
        m.NumMasked = ns; m.FoundState = 0;
        m.rHigh = m.rScale; return;
    }
    const uint32_t fFound = FS_UNI(fs_readlane(ls.sf >> 8, k)), succ = FS_UNI(fs_readlane(ls.succ, k));
    m.fsSym = (uint32_t)symbol; m.fsSucc = succ;
    prefetch_successor(m, c, succ);
    if (FS_UNI(base + k) == 0) {                              // most probable symbol
        m.PrevSuccess = (2 * (m.rHigh = fFound) > m.rScale);
        m.FoundState = p; m.fsFreq = fFound + 4; S_FREQ_SET(p, fFound + 4); C_SF_SET(c, m.rScale + 4);
        mc.set_sf(m.rScale + 4);
        if (fFound + 4 > MAX_FREQ) { rescale(m, c); fs_reload(m); mc = ctx_load(m, c); }
        m.rLow = 0; return;
    }
    m.PrevSuccess = 0;
    m.rHigh = (m.rLow = LoCnt) + fFound;
    // update1: +4, then bubble one position up if it now outweighs its predecessor
    m.FoundState = p; m.fsFreq = fFound + 4; C_SF_SET(c, m.rScale + 4);
    mc.set_sf(m.rScale + 4);
    uint32_t prevSf, prevSucc;
    if (k > 0) { prevSf = FS_UNI(fs_readlane(ls.sf, k - 1)); prevSucc = FS_UNI(fs_readlane(ls.succ, k - 1)); }
    else { const St t = st_load(m, p - 6); prevSf = t.sym | (t.freq << 8); prevSucc = t.succ; }
    if (fFound + 4 > (prevSf >> 8)) {
        state_store(m, p - 6, (uint32_t)symbol | ((fFound + 4) << 8), succ);
        state_store(m, p, prevSf, prevSucc);
        m.FoundState = p - 6;
        if (fFound + 4 > MAX_FREQ) { rescale(m, c); fs_reload(m); mc = ctx_load(m, c); }
    } else S_FREQ_SET(p, fFound + 4);
}

FS_DEV void encodeSymbol2(Coder& m, uint32_t c, Ctx& mc, int symbol, Ctx& sufRec, uint32_t& sufCtx)
{
    FS_REGION(8);
    const uint32_t nsC = mc.ns(), stats = mc.w1;
    // NumStats of the suffix context, in flight together with the state fetch below (the root has no suffix and,
    // with all 256 symbols, never needs it: Model.cpp:509)
    CtxRaw sr; sr.a = sr.b = sr.d = 0;
    if (mc.suff) sr = ctx_issue(m, mc.suff);
    LaneStates ls = lane_states(m, stats, nsC, 0);
    sufCtx = mc.suff;
    if (mc.suff) sufRec = ctx_finish(sr);
    // makeEscFreq2
    uint32_t seeIdx = 0xFFFFFFFFu, see = 0;
    if (nsC != 0xFF) {
        const uint32_t sufNs = mc.suff ? sufRec.ns() : 0u;
        seeIdx = (FS_UNI(m.sh->QT[nsC + 3]) - 4u) * 32u + (mc.sf() > 10u * (nsC + 1u)) + 2u * (2u * nsC < sufNs + m.NumMasked) + mc.flags();
        see = FS_LDS_RD(m.sh->SEE2[seeIdx]);
        const uint32_t shift = (see >> 16) & 0xFFu; uint32_t summ = see & 0xFFFFu;
        const uint32_t r = summ >> shift; summ = (summ - r) & 0xFFFFu;
        see = (see & 0xFFFF0000u) | summ;
        m.rScale = r + !r;
    } else m.rScale = 1;
    // unmasked states in list order, 64 per step
    const uint8_t esc = (uint8_t)m.EscCount;
    uint32_t LoCnt = 0, p = 0, fFound = 0, succ = 0, tail = 0; bool found = false;
#if FS_WIDE
    if (FS_UNI(nsC) < FS_WAVE) {                             // the whole list is in the lanes: no loop
        const uint32_t sy = ls.sf & 0xFFu;
        const bool unmasked = ls.valid && m.sh->CharMask[sy] != esc;
        const uint64_t hit = fs_ballot(unmasked && (int)sy == symbol);
        const uint32_t all = fs_wave_sum8(ls.sf >> 8, unmasked);
        if (hit != 0) {
            const uint32_t k = FS_UNI(fs_ctz64(hit));
            LoCnt = k != 0 ? fs_wave_sum8(ls.sf >> 8, unmasked && (uint32_t)FS_LANE() < k) : 0u;
            fFound = FS_UNI(fs_readlane(ls.sf >> 8, k)); succ = FS_UNI(fs_readlane(ls.succ, k));
            tail = all - LoCnt - fFound;
            if (unmasked && (uint32_t)FS_LANE() <= k) m.sh->CharMask[sy] = esc;     // visited states are marked
            p = stats + 6u * k; found = true;
        } else {
            LoCnt = all;
            if (unmasked) m.sh->CharMask[sy] = esc;
        }
    } else
#endif
    for (uint32_t base = 0;;) {
        const uint32_t sy = ls.sf & 0xFFu;
        const bool unmasked = ls.valid && m.sh->CharMask[sy] != esc;
        if (found) tail += fs_wave_sum8(ls.sf >> 8, unmasked);
        else {
            const uint64_t hit = fs_ballot(unmasked && (int)sy == symbol);
            if (hit) {
                const uint32_t k = fs_ctz64(hit);
                LoCnt += fs_wave_sum8(ls.sf >> 8, unmasked && (uint32_t)FS_LANE() < k);
                tail += fs_wave_sum8(ls.sf >> 8, unmasked && (uint32_t)FS_LANE() > k);
                fFound = FS_UNI(fs_readlane(ls.sf >> 8, k)); succ = FS_UNI(fs_readlane(ls.succ, k));
                if (unmasked && (uint32_t)FS_LANE() <= k) m.sh->CharMask[sy] = esc;     // visited states are marked
                p = stats + 6u * (base + k); found = true;
            } else {
                LoCnt += fs_wave_sum8(ls.sf >> 8, unmasked);
                if (unmasked) m.sh->CharMask[sy] = esc;
            }
        }
        if (base + FS_WAVE > nsC) break;
        base += FS_WAVE; ls = lane_states(m, stats, nsC, base);
    }
    FS_WAVE_SYNC();
    if (FS_UNI((uint32_t)found) == 0) {
        m.rHigh = (m.rScale += (m.rLow = LoCnt));
        if (seeIdx != 0xFFFFFFFFu) m.sh->SEE2[seeIdx] = (see & 0xFFFF0000u) | ((see + m.rScale) & 0xFFFFu);
        m.NumMasked = nsC;
        return;
    }
    m.rLow = LoCnt; m.rHigh = LoCnt + fFound;
    m.rScale += LoCnt + fFound + tail;
    if (seeIdx != 0xFFFFFFFFu) {                                   // psee2c->update()
        uint32_t summ = see & 0xFFFFu, shift = (see >> 16) & 0xFFu, count = (see >> 24) & 0xFFu;
        count = (count - 1) & 0xFFu;
        if (count == 0) {
            uint32_t kk = summ >> shift;
            kk = PERIOD_BITS - (kk > 40) - (kk > 280) - (kk > 1020);
            if (kk < shift) { summ >>= 1; shift--; }
            else if (kk > shift) { summ = (summ << 1) & 0xFFFFu; shift++; }
            count = (6u << shift) & 0xFFu;
        }
        m.sh->SEE2[seeIdx] = summ | (shift << 16) | (count << 24);
    }
    // update2
    m.FoundState = p; m.fsSym = (uint32_t)symbol; m.fsFreq = fFound + 4; m.fsSucc = succ;
    S_FREQ_SET(p, fFound + 4); C_SF_SET(c, mc.sf() + 4);
    mc.set_sf(mc.sf() + 4);
    if (fFound + 4 > MAX_FREQ) { rescale(m, c); fs_reload(m); mc = ctx_load(m, c); }
    m.EscCount = (m.EscCount + 1) & 0xFFu; m.RunLength = INIT_RL;
}

#if FS_WIDE
#include "ppmd_window.h"
#endif


// Encode one member.  `arena` = ARENA_BYTES of 16-byte aligned scratch (content irrelevant),
// returns the member size (clipped at outCap like the reference's ByteStream::Put).
// queued (64-lane builds): the caller runs coder_wave() on a second wavefront of the workgroup; the member's size then
// goes to *sizeOut from there and the return value is 0.
FS_DEV uint32_t encode_member(fs_gptr arena, FS_LDS Shared* sh, fs_cgptr in, uint32_t n, fs_gptr out, uint32_t outCap,
                              uint32_t* restartsOut, bool queued = false, FS_GLOBAL uint32_t* sizeOut = nullptr, uint32_t qTail = 0, uint32_t* qTailOut = nullptr)
{
    Coder m;
    m.ldsHeads = queued ? 1u : 0u;
    m.hb = arena - 1; m.sh = sh; m.out = out; m.outCap = outCap; m.outPos = 0; sh->restarts = 0;
    // (nothing is assumed about how far the coder wave has come with the stream before: the first push looks.  `qHeadSeen = qTail`
    // -- "the ring is empty" -- held in practice, the model's start-up outlasts any backlog, but was a promise nobody made:
    // found on the lock-step emulation at the end of round 3, where a range-coded stream's last windows were still in the ring)
    m.queued = queued ? 1u : 0u; m.qTail = qTail; m.qHeadSeen = qTail - CQ_SIZE;
    m.inAhead = 0; m.newCtx = 0;
    m.NumMasked = 0; m.FoundState = 0; m.BSumm = 0; m.rLow = m.rHigh = m.rScale = 0; m.fsSym = m.fsFreq = m.fsSucc = 0;
    for (uint32_t i = (uint32_t)FS_LANE(); i < 16u; i += FS_WAVE) sh->winStats[i] = 0u;
#if defined(FS_SER_PROFILE)
    for (uint32_t i = (uint32_t)FS_LANE(); i < 8u; i += FS_WAVE) sh->serStats[i] = 0u;
#endif
    for (uint32_t i = (uint32_t)FS_LANE(); i < 260u; i += FS_WAVE) sh->QT[i] = (uint8_t)QTable(i);
    // zero the 64-byte guard behind the heap: GlueFreeBlocks may read one stamp past the end
    for (uint32_t i = (uint32_t)FS_LANE(); i < 16u; i += FS_WAVE) *(fs_gptr32)(arena + SA_SIZE + 4u * i) = 0u;
    FS_WAVE_SYNC();
#if FS_WIDE
    if (m.queued) {       // hand the coder wave this stream's output buffer: box (s & 1), free once stream s - 2 has been opened
        const uint32_t s = FS_UNI(FS_LDS_RD(sh->qOpened));
        if (s >= 2u) cq_wait_starts(m, s - 1u);
        if (FS_LANE() == 0) {
            const uint64_t o = (uint64_t)(uintptr_t)out, z = (uint64_t)(uintptr_t)sizeOut;
            FS_LDS uint32_t* box = sh->qBox[s & 1u];
            box[0] = (uint32_t)o; box[1] = (uint32_t)(o >> 32); box[2] = outCap; box[3] = (uint32_t)z; box[4] = (uint32_t)(z >> 32);
            sh->qOpened = s + 1u;
        }
        FS_WAVE_SYNC();
        cq_push(m, CQ_CMD, CQ_START);          // (the release store of the tail publishes the box with the command)
    } else
#endif
    { put_byte(m, 0xCA); put_byte(m, MAX_ORDER); }
    m.low = 0; m.range = 0xFFFFFFFFu;
    StartModelRare(m);
    // input window: the next aligned dword is requested one step ahead of its first use
    const bool wide = (((uintptr_t)in) & 3u) == 0;
    uint32_t pos = 0, cur = 0, nxt = 0;
    if (wide && n >= 4) nxt = *(fs_cgptr32)in;
    Ctx mc; mc.a = mc.w1 = mc.suff = 0;
    uint32_t keep = 0, prevCtx = 0;
    Ctx sufRec = mc; uint32_t sufCtx = 0;
    m.pfCtx = 0; m.pf.a = m.pf.b = m.pf.d = 0;
#if FS_WIDE
    // windowed hit path: `hist` = the four bytes in front of `pos`; winSkip = serial symbols to code before the next attempt
    uint32_t hist = 0, winSkip = 0, winPenalty = 0;
    const bool windows = wide && n >= 64u;
#endif
    for (uint32_t MinContext = m.MaxContext;;) {
#if FS_WIDE
        if (windows && FS_UNI((uint32_t)m.OrderFall) == 0u && FS_UNI(pos) >= 4u && FS_UNI(pos) < n) {
            hint_learn(m, FS_UNI(hist), FS_UNI(MinContext));
            // (a context CreateSuccessors has just made holds one symbol: no window starts in a binary context, and the attempt that
            // finds that out costs a fetch of 64 positions -- 9 000 of 73 000 attempts on a 3 M-symbol quality stream)
            const bool fresh = FS_UNI(m.newCtx) == FS_UNI(MinContext);
            m.newCtx = 0;
            if (FS_UNI(winSkip) != 0u) --winSkip;
            else if (!fresh) {
#if defined(FS_SER_PROFILE)
                uint64_t tW = FS_PROF_NOW();
#endif
                const uint32_t penaltyIfNone = winPenalty + 1u, skipIfNone = penaltyIfNone <= 3u ? 0u : (penaltyIfNone >= 9u ? 64u : 1u << (penaltyIfNone - 3u));
                const uint32_t done = window_step(m, in, n, FS_UNI(pos), FS_UNI(MinContext), hist);
#if defined(FS_SER_PROFILE)
                if (done == 0u) FS_PROF_ACC(m.sh->serStats[4], tW);
#endif
                // a window that stopped short did so in front of a symbol for the serial path: skip one attempt.  An attempt
                // that codes nothing doubles the pause (learning phase of a model, unpredictable streams).
                // (a failed attempt costs less than half a serial symbol, and the symbols behind a read boundary are plain hits
                // again after two or three: the first three failures in a row cost no pause at all -- round 5: 454 -> 449 ms on a
                // lone 3 M-symbol stream against a pause of one symbol --, then it grows)
                if (done != 0u) { winPenalty = 0; winSkip = done < 64u ? 1u : 0u; }
                else { ++winPenalty; winSkip = skipIfNone; }
                if (done != 0u) {
                    pos += done; MinContext = m.MaxContext; m.pfCtx = 0; keep = 0; prevCtx = 0;
                    if ((pos & 3u) != 0u && (pos | 3u) < n) { cur = *(fs_cgptr32)(in + (pos & ~3u)); if ((pos & ~3u) + 8u <= n) nxt = *(fs_cgptr32)(in + (pos & ~3u) + 4u); }
                    else if ((pos | 3u) < n) nxt = *(fs_cgptr32)(in + pos);
                    continue;
                }
            }
        }
#endif
#if defined(FS_SER_PROFILE)
        uint64_t tS = FS_PROF_NOW();
        FS_STAT_ADD(m.sh->serStats[3], 1u);
#endif
        int c = -1;
        if (FS_UNI(pos) < n) {
            if (wide && (FS_UNI(pos) | 3u) < n) {
                if ((pos & 3u) == 0) { cur = FS_UNI(nxt); if (pos + 7u < n) nxt = *(fs_cgptr32)(in + pos + 4u); }
                c = (int)((cur >> (8u * (pos & 3u))) & 0xFFu);
            } else c = (int)fs_ld8(in + pos);
            pos++;
#if FS_WIDE
            hist = (hist >> 8) | ((uint32_t)c << 24);
#endif
        }
        // everything carried from symbol to symbol is wave-uniform by construction; saying so here keeps one value the
        // compiler could not prove uniform from turning the whole loop body into exec-masked (divergent) code
        MinContext = FS_UNI(MinContext); prevCtx = FS_UNI(prevCtx);
        mc.a = FS_UNI(mc.a); mc.w1 = FS_UNI(mc.w1); mc.suff = FS_UNI(mc.suff);
        m.OrderFall = (int32_t)FS_UNI(m.OrderFall); m.RunLength = (int32_t)FS_UNI(m.RunLength); m.pfCtx = FS_UNI(m.pfCtx);
        m.low = FS_UNI(m.low); m.range = FS_UNI(m.range); m.NumMasked = FS_UNI(m.NumMasked); m.EscCount = FS_UNI(m.EscCount); m.PrevSuccess = FS_UNI(m.PrevSuccess);
        m.pText = FS_UNI(m.pText); m.UnitsStart = FS_UNI(m.UnitsStart); m.LoUnit = FS_UNI(m.LoUnit); m.HiUnit = FS_UNI(m.HiUnit); m.outPos = FS_UNI(m.outPos);
        m.MaxContext = FS_UNI(m.MaxContext); m.BSumm = (int32_t)FS_UNI(m.BSumm);
        // first context of the symbol: still in registers (a context that succeeded itself), requested during the
        // previous symbol, or fetched now
        if (m.pfCtx == MinContext) mc = ctx_finish(m.pf);
        else if (!(FS_UNI(keep) && MinContext == prevCtx)) mc = ctx_load(m, MinContext);
        m.pfCtx = 0; keep = 0; prevCtx = MinContext; sufCtx = 0;
        mc.a = FS_UNI(mc.a); mc.w1 = FS_UNI(mc.w1); mc.suff = FS_UNI(mc.suff);
        FS_PATH(g_path[0]);
#if defined(FS_SER_PROFILE)
        FS_PROF_ACC(m.sh->serStats[0], tS);
#endif
        if (mc.ns() != 0) { FS_PATH(g_path[2]); encodeSymbol1(m, MinContext, mc, c); rc_encode(m); if (m.FoundState) { if (m.rLow == 0) FS_PATH(g_path[3]); else FS_PATH(g_path[4]); } }
        else { FS_PATH(g_path[1]); encodeBinSymbol(m, MinContext, mc, c, sufRec, sufCtx); }
#if defined(FS_SER_PROFILE)
        FS_PROF_ACC(m.sh->serStats[1], tS);
#endif
        uint32_t stop = 0;
        uint64_t tSer = FS_PROF_NOW();
        while (FS_UNI(m.FoundState) == 0) {
            rc_normalize(m);
            do {
                if (FS_UNI(mc.suff) == 0) { stop = 1; break; }
                m.OrderFall++; MinContext = mc.suff;
                if (FS_UNI(sufCtx) == FS_UNI(MinContext)) mc = sufRec; else mc = ctx_load(m, MinContext);
                sufCtx = 0;
            } while (FS_UNI(mc.ns()) == FS_UNI(m.NumMasked));
            if (FS_UNI(stop)) break;
            FS_PATH(g_path[5]);
            encodeSymbol2(m, MinContext, mc, c, sufRec, sufCtx); rc_encode(m);
            if (m.FoundState) FS_PATH(g_path[6]);
        }
        if (FS_UNI(stop)) break;
        FS_PROF_ACC(m.sh->winStats[6], tSer);                          // escapes: suffix walk + encodeSymbol2 rounds
        const uint32_t succ = FS_UNI(m.fsSucc);
        FS_SYMHOOK(prevCtx, MinContext, mc, m, succ);
        if (FS_UNI((uint32_t)m.OrderFall) == 0 && succ >= FS_UNI(m.UnitsStart)) { FS_PATH(g_path[7]); m.MaxContext = succ; keep = (succ == MinContext) ? 1u : 0u; }
        else { UpdateModel(m, MinContext, mc, FS_UNI(sufCtx) != 0 && FS_UNI(sufCtx) == FS_UNI(mc.suff), sufRec); if (FS_UNI(m.EscCount) == 0) clear_mask(m); FS_PROF_ACC(m.sh->winStats[7], tSer); }
#if defined(FS_SER_PROFILE)
        uint64_t tE = FS_PROF_NOW();
#endif
        rc_normalize(m); MinContext = m.MaxContext;
#if defined(FS_SER_PROFILE)
        FS_PROF_ACC(m.sh->serStats[2], tE);
#endif
    }
    if (restartsOut) *restartsOut = FS_UNI(m.sh->restarts);
    if (m.queued) { cq_push(m, CQ_CMD, CQ_END); if (qTailOut) *qTailOut = m.qTail; return 0u; }
    for (int i = 0; i < 4; i++) { put_byte(m, m.low >> 24); m.low <<= 8; }
    return m.outPos;
}

#if FS_WIDE
// The coder wave of the two-wave form: works the entries of the ring off in order (see the range-coder section).  Runs
// until the exit command; every wait is for the model wave, which never waits for anything but ring space.
// RC: the wave also owns the range coder of rc_core.h (64-bit low) for the range-coded streams whose triples come through the ring
// (fs_encode_streams2_w); the kernel every lossless launch takes instantiates coder_wave<false> and carries none of it.
// QV: what the wave does with a QVZ stream's entries (fsqvz::WaveCoder: the arithmetic coder of arith.cpp over them; NoQvz: no such streams)
template <bool RC, class QV = NoQvz> FS_DEV void coder_wave(FS_LDS Shared* sh)
{
    Coder m;                                                   // only the coder's own fields are used here
    m.sh = sh; m.queued = 0; m.low = 0; m.range = 0xFFFFFFFFu; m.out = nullptr; m.outCap = 0; m.outPos = 0;
    FS_GLOBAL uint32_t* sizeOut = nullptr;
    const uint32_t lane = (uint32_t)FS_LANE();
    uint32_t head = 0, starts = 0;
    uint64_t rcLow = 0; uint32_t rcRange = 0xFFFFFFFFu; bool rcMode = false;      // (RC) the range coder of rc_core.h (64-bit low) while a range-coded stream is open
    typename QV::State qs; bool qvMode = false;                                    // (QV) the arithmetic coder while a QVZ stream is open
    for (;;) {
        uint32_t tail;
        for (;;) { tail = FS_Q_LOAD(sh->qTail); if (tail != head) break; FS_Q_IDLE(); }
        const uint32_t n = tail - head < (uint32_t)FS_WAVE ? tail - head : (uint32_t)FS_WAVE;
        uint32_t eA = 0, eM = 0;
        if (lane < n) { eA = sh->qA[(head + lane) & (CQ_SIZE - 1u)]; eM = sh->qM[(head + lane) & (CQ_SIZE - 1u)]; }
        head += n;
        FS_Q_STORE(sh->qHead, head);                            // the batch is in registers: its slots are free again
        const bool special = lane < n && (eA >> 31) != 0u;
        uint32_t eL = 1u, eX = 0u, eY = 0u;
        if (QV::on && lane < n && !special && (eM & CQ_QVZ) != 0u) {          // a QVZ symbol: its two fractions of the interval, all the batch's at once
            uint32_t a0, a1, a2, a3;
            QV::prepare(eA, eM, a0, a1, a2, a3);
            eA = a0; eM = a1; eX = a2; eY = a3;
        } else if (RC && lane < n && !special && (eA & CQ_RC) != 0u) {       // a range-coded stream's triple: frequency | cumulative frequency, reciprocal of the total
            const uint32_t F = ((eA >> 16) & 0x3FFFu) | (((eM >> 16) & 3u) << 14), LO = eA & 0xFFFFu;
            const Recip rc = recip_make(eM & 0xFFFFu);
            eA = LO | (F << 16); eM = rc.mul; eL = rc.l;
        } else if (lane < n && !special) {                             // all the batch's reciprocals at once
            const Recip rc = recip_make(eM);
            eA = (eA & 0x00FFFFFFu) | ((rc.l - 1u) << 24); eM = rc.mul;
        }
        uint64_t todo = fs_ballot(special);
        m.low = FS_UNI(m.low); m.range = FS_UNI(m.range); m.outPos = FS_UNI(m.outPos);
        for (uint32_t i = 0; i < n;) {
            const uint64_t ahead = todo >> i;
            const uint32_t stop = ahead ? i + fs_ctz64(ahead) : n;
            if (QV::on && qvMode) {      // arithmetic_encoder_step (arith.cpp:33-103) over the symbols [i, stop)
                for (; i < stop; ++i) QV::code(qs, FS_UNI(fs_readlane(eA, i)), FS_UNI(fs_readlane(eM, i)), FS_UNI(fs_readlane(eX, i)), FS_UNI(fs_readlane(eY, i)));
            }
            if (RC && rcMode) {      // RangeEncoder::EncodeFrequency (rc/RangeCoder.h:40-84) over the triples [i, stop)
                for (; i < stop; ++i) {
                    const uint32_t FL = FS_UNI(fs_readlane(eA, i)), M0 = FS_UNI(fs_readlane(eM, i)), L0 = FS_UNI(fs_readlane(eL, i));
                    rcRange = recip_div(rcRange, M0, L0);
                    rcLow += (uint32_t)(rcRange * (FL & 0xFFFFu));
                    rcRange *= FL >> 16;
                    while (rcRange <= 0x00ffffffu) {
                        if ((rcLow ^ (rcLow + rcRange)) & 0xff00000000000000ULL) { const uint32_t x = (uint32_t)rcLow; rcRange = (x | 0x00ffffffu) - x; }
                        rc_put(m, (uint32_t)(rcLow >> 56));
                        rcLow <<= 8; rcRange <<= 8;
                    }
                }
            }
            // plain hits [i, stop): Coder.hpp:13-17 with the division by multiplication, then Model.cpp:580
            #define FS_CODE_ONE(A_, M_) do { \
                const uint32_t t_ = fs_mulhi(m.range, M_), rr_ = (t_ + ((m.range - t_) >> 1)) >> (A_ >> 24); \
                m.low += (A_ & 0xFFFFu) * rr_; m.range = rr_ * ((A_ >> 16) & 0x7Fu); \
                if (__builtin_expect(m.range < TOP, 0)) rc_shift_out(m); } while (0)
            while (i + 4u <= stop) {
                const uint32_t A0 = FS_UNI(fs_readlane(eA, i)), M0 = FS_UNI(fs_readlane(eM, i)), A1 = FS_UNI(fs_readlane(eA, i + 1u)), M1 = FS_UNI(fs_readlane(eM, i + 1u));
                const uint32_t A2 = FS_UNI(fs_readlane(eA, i + 2u)), M2 = FS_UNI(fs_readlane(eM, i + 2u)), A3 = FS_UNI(fs_readlane(eA, i + 3u)), M3 = FS_UNI(fs_readlane(eM, i + 3u));
                FS_CODE_ONE(A0, M0); FS_CODE_ONE(A1, M1); FS_CODE_ONE(A2, M2); FS_CODE_ONE(A3, M3);
                i += 4u;
            }
            for (; i < stop; ++i) { const uint32_t A0 = FS_UNI(fs_readlane(eA, i)), M0 = FS_UNI(fs_readlane(eM, i)); FS_CODE_ONE(A0, M0); }
            #undef FS_CODE_ONE
            if (i >= n) break;
            const uint32_t A = FS_UNI(fs_readlane(eA, i)), M = FS_UNI(fs_readlane(eM, i));
            ++i;
            if (A != CQ_CMD) {                                  // a step of the serial path
                const uint32_t rr = m.range / (M >> 16);
                m.low += (A & 0xFFFFu) * rr; m.range = rr * (M & 0xFFFFu);
                rc_shift_out(m);
            } else if (M == CQ_START) {
                FS_LDS uint32_t* box = sh->qBox[starts & 1u];
                const uint64_t o = (uint64_t)FS_LDS_RD(box[0]) | ((uint64_t)FS_LDS_RD(box[1]) << 32), z = (uint64_t)FS_LDS_RD(box[3]) | ((uint64_t)FS_LDS_RD(box[4]) << 32);
                m.out = (fs_gptr)(uintptr_t)o; sizeOut = (FS_GLOBAL uint32_t*)(uintptr_t)z; m.outCap = FS_LDS_RD(box[2]); m.outPos = 0;
                ++starts; FS_Q_STORE(sh->qStarts, starts);            // the box is read: the model wave may use it for the stream after next
                put_byte(m, 0xCA); put_byte(m, MAX_ORDER);
                m.low = 0; m.range = 0xFFFFFFFFu;
            } else if (M == CQ_END) {
                for (int k = 0; k < 4; k++) { put_byte(m, m.low >> 24); m.low <<= 8; }
                if (sizeOut) *sizeOut = m.outPos;
            } else if (RC && M == CQ_START_RC) {
                FS_LDS uint32_t* box = sh->qBox[starts & 1u];
                const uint64_t o = (uint64_t)FS_LDS_RD(box[0]) | ((uint64_t)FS_LDS_RD(box[1]) << 32), z = (uint64_t)FS_LDS_RD(box[3]) | ((uint64_t)FS_LDS_RD(box[4]) << 32);
                m.out = (fs_gptr)(uintptr_t)o; sizeOut = (FS_GLOBAL uint32_t*)(uintptr_t)z; m.outCap = FS_LDS_RD(box[2]); m.outPos = 0;
                cq_take_priority(FS_LDS_RD(box[5]));
                ++starts; FS_Q_STORE(sh->qStarts, starts);
                rcLow = 0; rcRange = 0xFFFFFFFFu; rcMode = true;
            } else if (RC && (M == CQ_END_RC || M == CQ_END_RC_BAD)) {
                for (int k = 0; k < 8; k++) { rc_put(m, (uint32_t)(rcLow >> 56)); rcLow <<= 8; }      // TEncoder::End: eight flush bytes
                if (sizeOut) *sizeOut = M == CQ_END_RC ? m.outPos : 0xFFFFFFFFu;
                rcMode = false; cq_take_priority(0u);
            } else if (QV::on && M == CQ_START_QVZ) {
                FS_LDS uint32_t* box = sh->qBox[starts & 1u];
                const uint64_t o = (uint64_t)FS_LDS_RD(box[0]) | ((uint64_t)FS_LDS_RD(box[1]) << 32), z = (uint64_t)FS_LDS_RD(box[3]) | ((uint64_t)FS_LDS_RD(box[4]) << 32);
                sizeOut = (FS_GLOBAL uint32_t*)(uintptr_t)z;
                QV::start(qs, (fs_gptr)(uintptr_t)o, FS_LDS_RD(box[2]));
                cq_take_priority(FS_LDS_RD(box[5]));
                ++starts; FS_Q_STORE(sh->qStarts, starts);
                qvMode = true;
            } else if (QV::on && (M == CQ_END_QVZ || M == CQ_END_QVZ_BAD)) {
                const uint32_t size = QV::finish(qs);
                if (sizeOut) *sizeOut = M == CQ_END_QVZ ? size : 0xFFFFFFFFu;
                qvMode = false; cq_take_priority(0u);
            } else return;                                      // CQ_EXIT
        }
    }
}
#endif

#undef HP
#undef BL
}  // namespace fsppmd
