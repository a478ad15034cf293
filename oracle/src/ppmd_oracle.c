/* TEST INFRASTRUCTURE (oracle) -- CPU restatement, never linked into the product.
 *
 * PPMd var.J (PPMII) encoder exactly as fastore_pack drives it: order 4, 16 MiB
 * sub-allocator, no order cut-off, model restarted per member, member = 0xCA 0x04 <coder
 * bytes> <4 flush bytes>  (reference: fastore/ppmd/PPMd.cpp:119-154, Model.cpp:109-140,
 * 212-586, SubAlloc.hpp, Coder.hpp; call site fastore_pack/FastqCompressor.cpp:1096-1118).
 *
 * Restated in plain C over one flat byte heap addressed by 32-bit indices
 * (index = byte offset + 1, 0 = NULL -- the reference's HeapNull = HeapStart-1 convention,
 * SubAlloc.hpp:35-37,157), so that every pointer comparison of the reference
 * ("succ >= UnitsStart", "iSuccessor <= iUpBranch") is the same integer comparison here.
 * Because CutOff is FALSE on this path, RestoreModelRare always takes the full-restart branch
 * (Model.cpp:188-192); cutOff/ExpandTextArea/MoveUnitsUp are unreachable and not restated.
 *
 * Parity pin: tests/test_oracle_ppmd.py checks this file against the real reference
 * (oracle/_ref/ref_driver ppmd) and against committed golden vectors.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { UNIT_SIZE = 12, N1 = 4, N2 = 4, N3 = 4, N4 = (128 + 3 - 1 * N1 - 2 * N2 - 3 * N3) / 4,
       N_INDEXES = N1 + N2 + N3 + N4 };
enum { UP_FREQ = 5, INT_BITS = 7, PERIOD_BITS = 7, TOT_BITS = INT_BITS + PERIOD_BITS,
       INTERVAL = 1 << INT_BITS, BIN_SCALE = 1 << TOT_BITS, ROUND = 16, MAX_FREQ = 124 };
enum { TOP = 1 << 24, BOT = 1 << 15 };
enum { SA_SIZE = 16 << 20, MAX_ORDER = 4 };

typedef struct { uint16_t Summ; uint8_t Shift, Count; } see2_t;

typedef struct {
    uint8_t* heap;              /* SA_SIZE + slack + list heads                     */
    uint32_t pText, UnitsStart, LoUnit, HiUnit;   /* heap indices (offset+1)       */
    uint32_t BList;             /* index of BList[0]; node i at BList + 12*i        */
    uint32_t GlueCount, GlueCount1;
    uint8_t Indx2Units[N_INDEXES], Units2Indx[128], NS2BSIndx[256], QTable[260];
    see2_t SEE2Cont[23][32], DummySEE2Cont;
    uint16_t BinSumm[25][64];
    uint8_t CharMask[256], NumMasked, PrevSuccess, EscCount;
    int BSumm, OrderFall, RunLength, InitRL;
    uint32_t MaxContext, FoundState;              /* heap indices                   */
    uint32_t low, range;                          /* Coder.hpp:7-28                 */
    uint32_t rLow, rHigh, rScale;                 /* SUBRANGE Range                 */
    uint8_t* out; size_t outCap, outPos;
    uint64_t restarts;
} ppmd_t;

/* ---- raw heap access (little endian, unaligned) ---- */
#define H(ix) (m->heap + ((ix) - 1u))
static inline uint16_t ld16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline void st16(uint8_t* p, uint16_t v) { memcpy(p, &v, 2); }
static inline void st32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }

/* PPM_CONTEXT (Model.cpp:30-58, pack(1)): NumStats@0 Flags@1 SummFreq@2 iStats@4 iSuffix@8;
   oneState() overlays SummFreq: Symbol@2 Freq@3 iSuccessor@4.  STATE: Symbol@0 Freq@1 iSucc@2 */
#define C_NS(c)      (H(c)[0])
#define C_FLAGS(c)   (H(c)[1])
#define C_SF(c)      ld16(H(c) + 2)
#define C_SF_SET(c,v) st16(H(c) + 2, (uint16_t)(v))
#define C_STATS(c)   ld32(H(c) + 4)
#define C_STATS_SET(c,v) st32(H(c) + 4, (v))
#define C_SUFF(c)    ld32(H(c) + 8)
#define C_SUFF_SET(c,v) st32(H(c) + 8, (v))
#define C_ONE(c)     ((c) + 2)              /* index of the embedded one-state        */
#define S_SYM(s)     (H(s)[0])
#define S_FREQ(s)    (H(s)[1])
#define S_SUCC(s)    ld32(H(s) + 2)
#define S_SUCC_SET(s,v) st32(H(s) + 2, (v))
/* BLK_NODE / MEM_BLK (SubAlloc.hpp:39-53): Stamp@0 NextIndx@4 NU@8 */
#define B_STAMP(b)   ld32(H(b))
#define B_STAMP_SET(b,v) st32(H(b), (v))
#define B_NEXT(b)    ld32(H(b) + 4)
#define B_NEXT_SET(b,v) st32(H(b) + 4, (v))
#define B_NU(b)      ld32(H(b) + 8)
#define B_NU_SET(b,v) st32(H(b) + 8, (v))
#define BL(i)        (m->BList + 12u * (uint32_t)(i))

static inline uint32_t U2B(uint32_t nu) { return 12u * nu; }

static inline void state_swap(ppmd_t* m, uint32_t a, uint32_t b)     /* Model.cpp:72-76 */
{ uint8_t t[6]; memcpy(t, H(a), 6); memcpy(H(a), H(b), 6); memcpy(H(b), t, 6); }
static inline void state_cpy(ppmd_t* m, uint32_t d, uint32_t s) { memmove(H(d), H(s), 6); }

/* ---- sub-allocator (SubAlloc.hpp) ---- */
static inline uint32_t blk_remove(ppmd_t* m, uint32_t n)            /* :55-58 */
{ uint32_t p = B_NEXT(n); B_NEXT_SET(n, B_NEXT(p)); B_STAMP_SET(n, B_STAMP(n) - 1); return p; }
static inline void blk_insert(ppmd_t* m, uint32_t n, uint32_t pv, uint32_t nu)   /* :59-63 */
{ B_NEXT_SET(pv, B_NEXT(n)); B_NEXT_SET(n, pv); B_STAMP_SET(pv, 0xFFFFFFFFu); B_NU_SET(pv, nu);
  B_STAMP_SET(n, B_STAMP(n) + 1); }
static inline int blk_avail(ppmd_t* m, uint32_t n) { return B_NEXT(n) != 0; }

static void SplitBlock(ppmd_t* m, uint32_t pv, uint32_t oldI, uint32_t newI)    /* :65-74 */
{
    uint32_t i, k, UDiff = m->Indx2Units[oldI] - m->Indx2Units[newI];
    uint32_t p = pv + U2B(m->Indx2Units[newI]);
    if (m->Indx2Units[i = m->Units2Indx[UDiff - 1]] != UDiff) {
        k = m->Indx2Units[--i]; blk_insert(m, BL(i), p, k);
        p += U2B(k); UDiff -= k;
    }
    blk_insert(m, BL(m->Units2Indx[UDiff - 1]), p, UDiff);
}

static void InitSubAllocator(ppmd_t* m)                              /* :98-107 */
{
    memset(H(m->BList), 0, 12 * (N_INDEXES + 1));
    m->pText = 1; m->HiUnit = 1 + SA_SIZE;
    uint32_t Diff = U2B(SA_SIZE / 8 / UNIT_SIZE * 7);
    m->LoUnit = m->UnitsStart = m->HiUnit - Diff; m->GlueCount = m->GlueCount1 = 0;
}

static void GlueFreeBlocks(ppmd_t* m)                                /* :108-133 */
{
    uint32_t i, k, sz, p, p0, p1;
    const uint32_t s0 = BL(N_INDEXES + 1);                            /* local MEM_BLK s0 */
    if (m->LoUnit != m->HiUnit) *H(m->LoUnit) = 0;
    p0 = s0; B_NEXT_SET(s0, 0); B_STAMP_SET(s0, 0); B_NU_SET(s0, 0);
    for (i = 0; i <= N_INDEXES; i++)
        while (blk_avail(m, BL(i))) {
            p = blk_remove(m, BL(i));
            if (!B_NU(p)) continue;
            while (B_STAMP(p1 = p + 12u * B_NU(p)) == 0xFFFFFFFFu) {
                B_NU_SET(p, B_NU(p) + B_NU(p1)); B_NU_SET(p1, 0);
            }
            B_NEXT_SET(p, B_NEXT(p0)); B_NEXT_SET(p0, p);             /* p0->link(p) */
            p0 = p;
        }
    while (blk_avail(m, s0)) {
        p = blk_remove(m, s0); sz = B_NU(p);
        if (!sz) continue;
        for (; sz > 128; sz -= 128, p += 12u * 128) blk_insert(m, BL(N_INDEXES - 1), p, 128);
        if (m->Indx2Units[i = m->Units2Indx[sz - 1]] != sz) {
            k = sz - m->Indx2Units[--i]; blk_insert(m, BL(k - 1), p + 12u * (sz - k), k);
        }
        blk_insert(m, BL(i), p, m->Indx2Units[i]);
    }
    m->GlueCount = 1u << (13 + m->GlueCount1++);
}

static uint32_t AllocUnitsRare(ppmd_t* m, uint32_t indx)             /* :134-150 */
{
    uint32_t i = indx;
    do {
        if (++i == N_INDEXES) {
            if (!m->GlueCount--) {
                GlueFreeBlocks(m);
                if (blk_avail(m, BL(i = indx))) return blk_remove(m, BL(i));
            } else {
                i = U2B(m->Indx2Units[indx]);
                return (m->UnitsStart - m->pText > i) ? (m->UnitsStart -= i) : 0;
            }
        }
    } while (!blk_avail(m, BL(i)));
    uint32_t r = blk_remove(m, BL(i)); SplitBlock(m, r, i, indx);
    return r;
}

static uint32_t AllocUnits(ppmd_t* m, uint32_t NU)                   /* :151-158 */
{
    uint32_t indx = m->Units2Indx[NU - 1];
    if (blk_avail(m, BL(indx))) return blk_remove(m, BL(indx));
    uint32_t r = m->LoUnit; m->LoUnit += U2B(m->Indx2Units[indx]);
    if (m->LoUnit <= m->HiUnit) return r;
    m->LoUnit -= U2B(m->Indx2Units[indx]); return AllocUnitsRare(m, indx);
}

static uint32_t AllocContext(ppmd_t* m)                              /* :159-163 */
{
    if (m->HiUnit != m->LoUnit) return (m->HiUnit -= UNIT_SIZE);
    return blk_avail(m, BL(0)) ? blk_remove(m, BL(0)) : AllocUnitsRare(m, 0);
}

static uint32_t ExpandUnits(ppmd_t* m, uint32_t oldPtr, uint32_t oldNU)   /* :177-184 */
{
    uint32_t i0 = m->Units2Indx[oldNU - 1], i1 = m->Units2Indx[oldNU - 1 + 1];
    if (i0 == i1) return oldPtr;
    uint32_t ptr = AllocUnits(m, oldNU + 1);
    if (ptr) { memcpy(H(ptr), H(oldPtr), 12u * oldNU); blk_insert(m, BL(i0), oldPtr, oldNU); }
    return ptr;
}

static uint32_t ShrinkUnits(ppmd_t* m, uint32_t oldPtr, uint32_t oldNU, uint32_t newNU) /* :185-195 */
{
    uint32_t i0 = m->Units2Indx[oldNU - 1], i1 = m->Units2Indx[newNU - 1];
    if (i0 == i1) return oldPtr;
    if (blk_avail(m, BL(i1))) {
        uint32_t ptr = blk_remove(m, BL(i1)); memcpy(H(ptr), H(oldPtr), 12u * newNU);
        blk_insert(m, BL(i0), oldPtr, m->Indx2Units[i0]);
        return ptr;
    }
    SplitBlock(m, oldPtr, i0, i1); return oldPtr;
}

static void FreeUnits(ppmd_t* m, uint32_t ptr, uint32_t NU)          /* :196-199 */
{ uint32_t indx = m->Units2Indx[NU - 1]; blk_insert(m, BL(indx), ptr, m->Indx2Units[indx]); }

/* ---- range coder (Coder.hpp) ---- */
static inline void put_byte(ppmd_t* m, uint32_t c)                   /* Stream.hpp:32-38 */
{ if (m->outPos < m->outCap) m->out[m->outPos++] = (uint8_t)c; }

static inline void rc_normalize(ppmd_t* m)                           /* Coder.hpp:11-17 */
{
    while ((m->low ^ (m->low + m->range)) < TOP ||
           (m->range < BOT && ((m->range = (0u - m->low) & (BOT - 1)), 1))) {
        put_byte(m, m->low >> 24);
        m->range <<= 8; m->low <<= 8;
    }
}
static inline void rc_encode(ppmd_t* m)                              /* Coder.hpp:18-21 */
{ m->low += m->rLow * (m->range /= m->rScale); m->range *= m->rHigh - m->rLow; }

/* ---- model ---- */
static void see2_init(see2_t* s, uint32_t v) { s->Summ = (uint16_t)(v << (s->Shift = PERIOD_BITS - 4)); s->Count = 7; }
static uint32_t see2_mean(see2_t* s) { uint32_t r = s->Summ >> s->Shift; s->Summ = (uint16_t)(s->Summ - r); return r + !r; }
static void see2_update(see2_t* s)                                   /* Model.cpp:26,79-86 */
{
    if (--s->Count == 0) {
        uint32_t i = s->Summ >> s->Shift;
        i = PERIOD_BITS - (i > 40) - (i > 280) - (i > 1020);
        if (i < s->Shift) { s->Summ >>= 1; s->Shift--; }
        else if (i > s->Shift) { s->Summ = (uint16_t)(s->Summ << 1); s->Shift++; }
        s->Count = (uint8_t)(6 << s->Shift);
    }
}

static void init_tables(ppmd_t* m)                                   /* Model.cpp:88-108 */
{
    uint32_t i, k, mm, Step;
    for (i = 0, k = 1; i < N1; i++, k += 1) m->Indx2Units[i] = (uint8_t)k;
    for (k++; i < N1 + N2; i++, k += 2) m->Indx2Units[i] = (uint8_t)k;
    for (k++; i < N1 + N2 + N3; i++, k += 3) m->Indx2Units[i] = (uint8_t)k;
    for (k++; i < N1 + N2 + N3 + N4; i++, k += 4) m->Indx2Units[i] = (uint8_t)k;
    for (k = i = 0; k < 128; k++) { i += (m->Indx2Units[i] < k + 1); m->Units2Indx[k] = (uint8_t)i; }
    m->NS2BSIndx[0] = 2 * 0; m->NS2BSIndx[1] = m->NS2BSIndx[2] = 2 * 1;
    memset(m->NS2BSIndx + 3, 2 * 2, 26); memset(m->NS2BSIndx + 29, 2 * 3, 256 - 29);
    for (i = 0; i < UP_FREQ; i++) m->QTable[i] = (uint8_t)i;
    for (mm = i = UP_FREQ, k = Step = 1; i < 260; i++) {
        m->QTable[i] = (uint8_t)mm;
        if (!--k) { k = ++Step; mm++; }
    }
}

static void StartModelRare(ppmd_t* m)                                /* Model.cpp:109-140 */
{
    static const signed char EscCoef[12] = {16, -10, 1, 51, 14, 89, 23, 35, 64, 26, -42, 43};
    int i, k, s; uint8_t i2f[25];
    memset(m->CharMask, 0, sizeof m->CharMask); m->EscCount = 1;
    m->OrderFall = MAX_ORDER;
    InitSubAllocator(m);
    m->RunLength = m->InitRL = -((MAX_ORDER < 13) ? MAX_ORDER : 13);
    m->MaxContext = AllocContext(m);
    C_NS(m->MaxContext) = 255; C_SF_SET(m->MaxContext, 255 + 2);
    C_STATS_SET(m->MaxContext, AllocUnits(m, 256 / 2));
    m->PrevSuccess = 0; C_SUFF_SET(m->MaxContext, 0); C_FLAGS(m->MaxContext) = 0;
    for (i = 0; i < 256; i++) {
        uint32_t st = C_STATS(m->MaxContext) + 6u * (uint32_t)i;
        S_SYM(st) = (uint8_t)i; S_FREQ(st) = 1; S_SUCC_SET(st, 0);
    }
    for (k = i = 0; i < 25; i2f[i++] = (uint8_t)(k + 1)) while (m->QTable[k] == i) k++;
    for (k = 0; k < 64; k++) {
        for (s = i = 0; i < 6; i++) s += EscCoef[2 * i + ((k >> i) & 1)];
        s = 128 * (s < 32 ? 32 : (s > 256 - 32 ? 256 - 32 : s));
        for (i = 0; i < 25; i++) m->BinSumm[i][k] = (uint16_t)(BIN_SCALE - s / i2f[i]);
    }
    for (i = 0; i < 23; i++) for (k = 0; k < 32; k++) see2_init(&m->SEE2Cont[i][k], 8 * i + 5);
}

static void RestoreModelRare(ppmd_t* m)                              /* Model.cpp:186-192 (!CutOff) */
{ m->pText = 1; StartModelRare(m); m->EscCount = 0; m->restarts++; }

static uint32_t CreateSuccessors(ppmd_t* m, int Skip, uint32_t p, uint32_t pc);

static uint32_t ReduceOrder(ppmd_t* m, uint32_t p, uint32_t pc)      /* Model.cpp:209-243 */
{
    uint32_t p1, pc1 = pc;
    uint32_t iUpBranch = m->pText; S_SUCC_SET(m->FoundState, iUpBranch);
    uint8_t tmp, sym = S_SYM(m->FoundState); m->OrderFall++;
    if (p) { pc = C_SUFF(pc); goto LOOP_ENTRY; }
    for (;;) {
        if (!C_SUFF(pc)) return pc;
        pc = C_SUFF(pc);
        if (C_NS(pc)) {
            if (S_SYM(p = C_STATS(pc)) != sym) do { tmp = S_SYM(p + 6); p += 6; } while (tmp != sym);
            tmp = (uint8_t)(2 * (S_FREQ(p) < MAX_FREQ - 3));
            S_FREQ(p) = (uint8_t)(S_FREQ(p) + tmp); C_SF_SET(pc, C_SF(pc) + tmp);
        } else { p = C_ONE(pc); S_FREQ(p) = (uint8_t)(S_FREQ(p) + (S_FREQ(p) < 11)); }
LOOP_ENTRY:
        if (S_SUCC(p)) break;
        S_SUCC_SET(p, iUpBranch); m->OrderFall++;
    }
    if (S_SUCC(p) <= iUpBranch) {
        p1 = m->FoundState; m->FoundState = p;
        S_SUCC_SET(p, CreateSuccessors(m, 0, 0, pc));
        m->FoundState = p1;
    }
    if (m->OrderFall == 1 && pc1 == m->MaxContext) {
        S_SUCC_SET(m->FoundState, S_SUCC(p));
        m->pText--;
    }
    return S_SUCC(p);
}

static void rescale(ppmd_t* m, uint32_t c)                           /* Model.cpp:244-281 */
{
    uint32_t f0, sf, EscFreq, a = (m->OrderFall != 0), i = C_NS(c);
    uint32_t p1, p; uint8_t tmp[6];
    C_FLAGS(c) &= 0x14;
    for (p = m->FoundState; p != C_STATS(c); p -= 6) state_swap(m, p, p - 6);
    f0 = S_FREQ(p); sf = C_SF(c);
    EscFreq = C_SF(c) - S_FREQ(p);
    S_FREQ(p) = (uint8_t)((S_FREQ(p) + a) >> 1); C_SF_SET(c, S_FREQ(p));
    do {
        p += 6; EscFreq -= S_FREQ(p);
        S_FREQ(p) = (uint8_t)((S_FREQ(p) + a) >> 1); C_SF_SET(c, C_SF(c) + S_FREQ(p));
        if (S_FREQ(p)) C_FLAGS(c) |= 0x08 * (S_SYM(p) >= 0x40);
        if (S_FREQ(p) > S_FREQ(p - 6)) {
            memcpy(tmp, H(p1 = p), 6);
            do { state_cpy(m, p1, p1 - 6); } while (tmp[1] > S_FREQ((p1 -= 6) - 6));
            memcpy(H(p1), tmp, 6);
        }
    } while (--i);
    if (S_FREQ(p) == 0) {
        do { i++; } while (S_FREQ(p -= 6) == 0);
        EscFreq += i; a = (C_NS(c) + 2u) >> 1;
        if ((C_NS(c) = (uint8_t)(C_NS(c) - i)) == 0) {
            memcpy(tmp, H(C_STATS(c)), 6); C_FLAGS(c) &= 0x18;
            tmp[1] = (uint8_t)((2u * tmp[1] + EscFreq - 1) / EscFreq);
            if (tmp[1] > MAX_FREQ / 3) tmp[1] = MAX_FREQ / 3;
            FreeUnits(m, C_STATS(c), a); memcpy(H(C_ONE(c)), tmp, 6);
            m->FoundState = C_ONE(c); return;
        }
        C_STATS_SET(c, ShrinkUnits(m, C_STATS(c), a, (C_NS(c) + 2u) >> 1));
    }
    C_SF_SET(c, C_SF(c) + ((EscFreq + 1) >> 1));
    if (m->OrderFall || (C_FLAGS(c) & 0x04) == 0) {
        a = (sf -= EscFreq) - f0;
        a = (f0 * C_SF(c) - sf * S_FREQ(C_STATS(c)) + a - 1) / a;
        a = a < 2u ? 2u : (a > MAX_FREQ / 2u - 18u ? MAX_FREQ / 2u - 18u : a);
    } else a = 2;
    m->FoundState = C_STATS(c);
    S_FREQ(m->FoundState) = (uint8_t)(S_FREQ(m->FoundState) + a); C_SF_SET(c, C_SF(c) + a);
    C_FLAGS(c) |= 0x04;
}

static uint32_t CreateSuccessors(ppmd_t* m, int Skip, uint32_t p, uint32_t pc)   /* Model.cpp:282-337 */
{
    uint8_t ct[12];                                                  /* PPM_CONTEXT ct */
    uint32_t iUpBranch = S_SUCC(m->FoundState);
    uint32_t ps[16], pps = 0;
    uint32_t cf, s0;
    uint8_t tmp, sym = S_SYM(m->FoundState);
    if (!Skip) {
        ps[pps++] = m->FoundState;
        if (!C_SUFF(pc)) goto NO_LOOP;
    }
    if (p) { pc = C_SUFF(pc); goto LOOP_ENTRY; }
    do {
        pc = C_SUFF(pc);
        if (C_NS(pc)) {
            if (S_SYM(p = C_STATS(pc)) != sym) do { tmp = S_SYM(p + 6); p += 6; } while (tmp != sym);
            tmp = (S_FREQ(p) < MAX_FREQ);
            S_FREQ(p) = (uint8_t)(S_FREQ(p) + tmp); C_SF_SET(pc, C_SF(pc) + tmp);
        } else {
            p = C_ONE(pc);
            S_FREQ(p) = (uint8_t)(S_FREQ(p) + ((!C_NS(C_SUFF(pc))) & (S_FREQ(p) < 11)));
        }
LOOP_ENTRY:
        if (S_SUCC(p) != iUpBranch) { pc = S_SUCC(p); break; }
        ps[pps++] = p;
    } while (C_SUFF(pc));
NO_LOOP:
    if (pps == 0) return pc;
    memset(ct, 0, sizeof ct);
    ct[0] = 0; ct[1] = (uint8_t)(0x10 * (sym >= 0x40));
    ct[2] = sym = *H(iUpBranch);                                      /* oneState().Symbol */
    st32(ct + 4, iUpBranch + 1);                                      /* oneState().iSuccessor */
    ct[1] |= 0x08 * (sym >= 0x40);
    if (C_NS(pc)) {
        if (S_SYM(p = C_STATS(pc)) != sym) do { tmp = S_SYM(p + 6); p += 6; } while (tmp != sym);
        s0 = C_SF(pc) - C_NS(pc) - (cf = S_FREQ(p) - 1u);
        cf = 1 + ((2 * cf <= s0) ? (12 * cf > s0) : ((cf + 2 * s0) / s0));
        ct[3] = (uint8_t)((cf < 7) ? cf : 7);
    } else ct[3] = S_FREQ(C_ONE(pc));
    do {
        uint32_t pc1 = AllocContext(m);
        if (!pc1) return 0;
        memcpy(H(pc1), ct, 8);
        C_SUFF_SET(pc1, pc);
        S_SUCC_SET(ps[--pps], pc = pc1);
    } while (pps != 0);
    return pc;
}

static const uint8_t ExpEscape[16] = {51, 43, 18, 12, 11, 9, 8, 7, 6, 5, 4, 3, 3, 2, 2, 2};

static void UpdateModel(ppmd_t* m, uint32_t MinContext)              /* Model.cpp:340-416 */
{
    uint8_t Flag, sym, FSymbol = S_SYM(m->FoundState);
    uint32_t ns1, ns, cf, sf, s0, FFreq = S_FREQ(m->FoundState);
    uint32_t iSuccessor, iFSuccessor = S_SUCC(m->FoundState);
    uint32_t pc, p = 0;
    if (C_SUFF(MinContext)) {
        pc = C_SUFF(MinContext);
        if (C_NS(pc)) {
            if (S_SYM(p = C_STATS(pc)) != FSymbol) {
                do { sym = S_SYM(p + 6); p += 6; } while (sym != FSymbol);
                if (S_FREQ(p) >= S_FREQ(p - 6)) { state_swap(m, p, p - 6); p -= 6; }
            }
            if (S_FREQ(p) < MAX_FREQ) {
                cf = 1 + (FFreq < 4 * 8);
                S_FREQ(p) = (uint8_t)(S_FREQ(p) + cf); C_SF_SET(pc, C_SF(pc) + cf);
            }
        } else { p = C_ONE(pc); S_FREQ(p) = (uint8_t)(S_FREQ(p) + (S_FREQ(p) < 11)); }
    }
    pc = m->MaxContext;
    if (!m->OrderFall && iFSuccessor) {
        S_SUCC_SET(m->FoundState, CreateSuccessors(m, 1, p, MinContext));
        if (!S_SUCC(m->FoundState)) goto RESTART_MODEL;
        m->MaxContext = S_SUCC(m->FoundState); return;
    }
    *H(m->pText) = FSymbol; m->pText++; iSuccessor = m->pText;
    if (m->pText >= m->UnitsStart) goto RESTART_MODEL;
    if (iFSuccessor) {
        if (iFSuccessor < m->UnitsStart) iFSuccessor = CreateSuccessors(m, 0, p, MinContext);
    } else iFSuccessor = ReduceOrder(m, p, MinContext);
    if (!iFSuccessor) goto RESTART_MODEL;
    if (!--m->OrderFall) { iSuccessor = iFSuccessor; m->pText -= (m->MaxContext != MinContext); }
    s0 = C_SF(MinContext) - FFreq; ns = C_NS(MinContext);
    Flag = (uint8_t)(0x08 * (FSymbol >= 0x40));
    for (; pc != MinContext; pc = C_SUFF(pc)) {
        if ((ns1 = C_NS(pc)) != 0) {
            if ((ns1 & 1) != 0) {
                p = ExpandUnits(m, C_STATS(pc), (ns1 + 1) >> 1);
                if (!p) goto RESTART_MODEL;
                C_STATS_SET(pc, p);
            }
            C_SF_SET(pc, C_SF(pc) + (m->QTable[ns + 4] >> 3));
        } else {
            p = AllocUnits(m, 1);
            if (!p) goto RESTART_MODEL;
            state_cpy(m, p, C_ONE(pc)); C_STATS_SET(pc, p);
            S_FREQ(p) = (uint8_t)((S_FREQ(p) <= MAX_FREQ / 3) ? (2 * S_FREQ(p) - 1) : (MAX_FREQ - 15));
            C_SF_SET(pc, S_FREQ(p) + (ns > 1) + ExpEscape[m->QTable[m->BSumm >> 8]]);
        }
        cf = 2 * FFreq * (C_SF(pc) + 4u); sf = s0 + C_SF(pc);
        if (cf <= 6 * sf) {
            cf = 1 + (cf > sf) + (cf > 3 * sf); C_SF_SET(pc, C_SF(pc) + 4);
        } else {
            cf = 4 + (cf > 8 * sf) + (cf > 10 * sf) + (cf > 13 * sf); C_SF_SET(pc, C_SF(pc) + cf);
        }
        C_NS(pc) = (uint8_t)(C_NS(pc) + 1);
        p = C_STATS(pc) + 6u * C_NS(pc); S_SUCC_SET(p, iSuccessor);
        S_SYM(p) = FSymbol; S_FREQ(p) = (uint8_t)cf;
        C_FLAGS(pc) |= Flag;
    }
    m->MaxContext = iFSuccessor;
    return;
RESTART_MODEL:
    RestoreModelRare(m);
}

static void encodeBinSymbol(ppmd_t* m, uint32_t c, int symbol)       /* Model.cpp:418-433 */
{
    uint32_t rs = C_ONE(c);
    uint16_t* bs = &m->BinSumm[m->QTable[S_FREQ(rs) - 1]][m->NS2BSIndx[C_NS(C_SUFF(c))] + m->PrevSuccess +
                                                          C_FLAGS(c) + ((m->RunLength >> 26) & 0x20)];
    m->BSumm = *bs;
    uint32_t tmp = (uint32_t)m->BSumm * (m->range >>= TOT_BITS);
    *bs = (uint16_t)(*bs - ((m->BSumm + ROUND) >> PERIOD_BITS));
    if (S_SYM(rs) == symbol) {
        *bs = (uint16_t)(*bs + INTERVAL); m->range = tmp;
        m->FoundState = rs; S_FREQ(rs) = (uint8_t)(S_FREQ(rs) + (S_FREQ(rs) < 196));
        m->RunLength++; m->PrevSuccess = 1;
    } else {
        m->low += tmp; m->range *= (uint32_t)(BIN_SCALE - m->BSumm);
        m->CharMask[S_SYM(rs)] = m->EscCount;
        m->NumMasked = m->PrevSuccess = 0; m->FoundState = 0;
    }
}

static void update1(ppmd_t* m, uint32_t c, uint32_t p)               /* Model.cpp:450-457 */
{
    m->FoundState = p; S_FREQ(p) = (uint8_t)(S_FREQ(p) + 4); C_SF_SET(c, C_SF(c) + 4);
    if (S_FREQ(p) > S_FREQ(p - 6)) {
        state_swap(m, p, p - 6); m->FoundState = (p -= 6);
        if (S_FREQ(p) > MAX_FREQ) rescale(m, c);
    }
}

static void encodeSymbol1(ppmd_t* m, uint32_t c, int symbol)         /* Model.cpp:458-481 */
{
    uint32_t p = C_STATS(c);
    uint32_t i = S_SYM(p), LoCnt = S_FREQ(p); m->rScale = C_SF(c);
    if ((int)i == symbol) {
        m->PrevSuccess = (2 * (m->rHigh = LoCnt) > m->rScale);
        m->FoundState = p; S_FREQ(p) = (uint8_t)(LoCnt += 4); C_SF_SET(c, C_SF(c) + 4);
        if (LoCnt > MAX_FREQ) rescale(m, c);
        m->rLow = 0; return;
    }
    i = C_NS(c); m->PrevSuccess = 0;
    while (S_SYM(p += 6) != symbol) {
        LoCnt += S_FREQ(p);
        if (--i == 0) {
            m->rLow = LoCnt; m->CharMask[S_SYM(p)] = m->EscCount;
            i = m->NumMasked = C_NS(c); m->FoundState = 0;
            do { p -= 6; m->CharMask[S_SYM(p)] = m->EscCount; } while (--i);
            m->rHigh = m->rScale; return;
        }
    }
    m->rHigh = (m->rLow = LoCnt) + S_FREQ(p); update1(m, c, p);
}

static void update2(ppmd_t* m, uint32_t c, uint32_t p)               /* Model.cpp:500-505 */
{
    m->FoundState = p; S_FREQ(p) = (uint8_t)(S_FREQ(p) + 4); C_SF_SET(c, C_SF(c) + 4);
    if (S_FREQ(p) > MAX_FREQ) rescale(m, c);
    m->EscCount++; m->RunLength = m->InitRL;
}

static see2_t* makeEscFreq2(ppmd_t* m, uint32_t c)                   /* Model.cpp:506-516 */
{
    see2_t* s;
    if (C_NS(c) != 0xFF) {
        s = m->SEE2Cont[m->QTable[C_NS(c) + 3] - 4] + (C_SF(c) > 10u * (C_NS(c) + 1u)) +
            2 * (2u * C_NS(c) < (uint32_t)C_NS(C_SUFF(c)) + m->NumMasked) + C_FLAGS(c);
        m->rScale = see2_mean(s);
    } else { s = &m->DummySEE2Cont; m->rScale = 1; }
    return s;
}

static void encodeSymbol2(ppmd_t* m, uint32_t c, int symbol)         /* Model.cpp:517-540 */
{
    see2_t* psee2c = makeEscFreq2(m, c);
    uint32_t Sym, LoCnt = 0, i = (uint32_t)C_NS(c) - m->NumMasked;
    uint32_t p1, p = C_STATS(c) - 6;
    do {
        do { Sym = S_SYM(p + 6); p += 6; } while (m->CharMask[Sym] == m->EscCount);
        m->CharMask[Sym] = m->EscCount;
        if ((int)Sym == symbol) goto SYMBOL_FOUND;
        LoCnt += S_FREQ(p);
    } while (--i);
    m->rHigh = (m->rScale += (m->rLow = LoCnt));
    psee2c->Summ = (uint16_t)(psee2c->Summ + m->rScale); m->NumMasked = C_NS(c);
    return;
SYMBOL_FOUND:
    m->rLow = LoCnt; m->rHigh = (LoCnt += S_FREQ(p));
    for (p1 = p; --i;) {
        do { Sym = S_SYM(p1 + 6); p1 += 6; } while (m->CharMask[Sym] == m->EscCount);
        LoCnt += S_FREQ(p1);
    }
    m->rScale += LoCnt;
    see2_update(psee2c); update2(m, c, p);
}

static void EncodeFile(ppmd_t* m, const uint8_t* in, size_t n)       /* Model.cpp:559-586 */
{
    size_t pos = 0;
    m->low = 0; m->range = 0xFFFFFFFFu;
    StartModelRare(m);
    for (uint32_t MinContext = m->MaxContext;;) {
        int c = (pos < n) ? in[pos++] : -1;
        if (C_NS(MinContext)) { encodeSymbol1(m, MinContext, c); rc_encode(m); }
        else encodeBinSymbol(m, MinContext, c);
        while (!m->FoundState) {
            rc_normalize(m);
            do {
                if (!C_SUFF(MinContext)) goto STOP_ENCODING;
                m->OrderFall++; MinContext = C_SUFF(MinContext);
            } while (C_NS(MinContext) == m->NumMasked);
            encodeSymbol2(m, MinContext, c); rc_encode(m);
        }
        if (!m->OrderFall && S_SUCC(m->FoundState) >= m->UnitsStart)
            m->MaxContext = S_SUCC(m->FoundState);
        else {
            UpdateModel(m, MinContext);
            if (m->EscCount == 0) { m->EscCount = 1; memset(m->CharMask, 0, sizeof m->CharMask); }
        }
        rc_normalize(m); MinContext = m->MaxContext;
    }
STOP_ENCODING:
    for (int i = 0; i < 4; i++) { put_byte(m, m->low >> 24); m->low <<= 8; }   /* Coder.hpp:22-27 */
}

/* Encode one PPMd member as LzCompressorSE::CompressBuffers does.  Returns the member size
 * (bytes written to out, clipped at cap exactly like ByteStream::Put).  restarts (optional)
 * receives the number of model restarts, used_hi (optional) the high-water mark of the unit
 * area (diagnostics for sizing the device arenas). */
size_t fso_ppmd_encode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, uint64_t* restarts)
{
    ppmd_t* m = (ppmd_t*)calloc(1, sizeof(ppmd_t));
    m->heap = (uint8_t*)calloc(1, (size_t)SA_SIZE + 64 + 12 * (N_INDEXES + 2));
    m->BList = 1 + SA_SIZE + 64;
    m->out = out; m->outCap = cap; m->outPos = 0;
    init_tables(m);
    put_byte(m, 0xCA); put_byte(m, MAX_ORDER);                        /* PPMd.cpp:131-135 */
    EncodeFile(m, in, n);
    size_t r = m->outPos;
    if (restarts) *restarts = m->restarts;
    free(m->heap); free(m);
    return r;
}
