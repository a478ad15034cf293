/* TEST INFRASTRUCTURE (oracle) -- CPU restatement, never linked into the product.
 *
 * The --lossy (QVZ) quality path of fastore_pack, for one block of reads:
 *   codebook in the footer       fastore/fastore_bin/QVZ.cpp:225-302  (ReadCodebook), layout BinFile.cpp:386-394
 *   alphabets                    fastore/fastore_pack/pmf.cpp:20-46, 322-396; quantizer.cpp:449-480
 *   WELL-1024a                   fastore/fastore_pack/well.cpp:16-57
 *   quantizer choice             fastore/fastore_pack/quantizer.cpp:522-531
 *   per-read loop                fastore/fastore_pack/FastqCompressor.cpp:318-364 (generator reset per block: 906-915)
 *   adaptive statistics          fastore/fastore_pack/qv_stream.cpp:19-71
 *   arithmetic coder             fastore/fastore_pack/arith.cpp:14-125 (m = 22)
 *   bit writer                   fastore/fastore_bin/BitMemory.h:251-313, 375-404
 * Parity pin: tests/test_oracle.py against vectors made by `oracle/_ref/ref_driver qvz` (the reference's own classes)
 * and against the quality streams inside the reference-made golden archive tests/golden/se_qvz.ref.cdata.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define A_SIZE 72
#define NOT_SYMBOL 127
#define INDEX_SLOTS (A_SIZE + 10)
#define NOT_FOUND 0xFFFFFFFFu

typedef struct { uint32_t size; uint8_t sym[2 * A_SIZE + 2]; uint32_t index[INDEX_SLOTS]; } alpha_t;
typedef struct { uint8_t q[A_SIZE]; alpha_t out; } quant_t;
typedef struct { uint32_t* counts; uint32_t card, step, n; } stats_t;
typedef struct { alpha_t in; uint8_t* qratio; quant_t* q; stats_t* st; } column_t;   /* q[2*idx] low, q[2*idx+1] high */

static int alpha_index(alpha_t* a)
{
    for (uint32_t i = 0; i < INDEX_SLOTS; ++i) a->index[i] = NOT_FOUND;
    for (uint32_t i = 0; i < a->size; ++i) { if (a->sym[i] >= INDEX_SLOTS) return -1; a->index[a->sym[i]] = i; }
    return 0;
}
static int alpha_union(const alpha_t* a, const alpha_t* b, alpha_t* r)
{
    alpha_t t; uint32_t i = 0, j = 0, k = 0;
    while (i < a->size && j < b->size) {
        if (a->sym[i] < b->sym[j]) t.sym[k++] = a->sym[i++];
        else if (a->sym[i] == b->sym[j]) { t.sym[k++] = a->sym[i]; i++; j++; }
        else t.sym[k++] = b->sym[j++];
        if (k > A_SIZE) return -1;
    }
    while (i < a->size) { if (k > A_SIZE) return -1; t.sym[k++] = a->sym[i++]; }
    while (j < b->size) { if (k > A_SIZE) return -1; t.sym[k++] = b->sym[j++]; }
    t.size = k;
    if (alpha_index(&t)) return -1;
    *r = t;
    return 0;
}
static int quant_load(quant_t* z, const uint8_t* line)
{
    uint32_t size = 0; uint8_t p;
    for (uint32_t i = 0; i < A_SIZE; ++i) z->q[i] = (uint8_t)(line[i] - 33);
    p = z->q[0];
    if (p != NOT_SYMBOL) z->out.sym[size++] = p;
    for (uint32_t x = 1; x < A_SIZE; ++x)
        if (z->q[x] != p) { p = z->q[x]; if (p == NOT_SYMBOL) break; z->out.sym[size++] = p; }
    z->out.size = size;
    return alpha_index(&z->out);
}

typedef struct { uint32_t state[32], n, bit_output, bits_left; } well_t;
static uint32_t well_next(well_t* w)
{
    uint32_t* s = w->state; uint32_t n = w->n;
    uint32_t z0 = s[(n + 31) & 31], v1 = s[(n + 3) & 31], v2 = s[(n + 24) & 31], v3 = s[(n + 10) & 31];
    uint32_t z1 = s[n] ^ (v1 ^ (v1 >> 8));
    uint32_t z2 = (v2 ^ (v2 << 19)) ^ (v3 ^ (v3 << 14));
    s[n] = z1 ^ z2;
    n = (n + 31) & 31;
    s[n] = (z0 ^ (z0 << 11)) ^ (z1 ^ (z1 << 7)) ^ (z2 ^ (z2 << 13));
    w->n = n;
    return s[n];
}
static uint32_t well_bits(well_t* w, uint32_t bits)
{
    uint32_t r;
    if (w->bits_left < bits) { w->bit_output = well_next(w); w->bits_left = 32; }
    r = w->bit_output & ((1u << bits) - 1);
    w->bit_output >>= bits; w->bits_left -= bits;
    return r;
}

/* MSB-first bit writer, as BitMemoryWriter behaves for PutBit / PutBits / FillLastByte / Flush */
typedef struct { uint8_t* out; size_t cap, pos; uint32_t cur, fill; } bitw_t;
static void bw_bit(bitw_t* w, uint32_t b)
{
    w->cur = (w->cur << 1) | (b & 1u);
    if (++w->fill == 8) { if (w->pos < w->cap) w->out[w->pos] = (uint8_t)w->cur; w->pos++; w->cur = 0; w->fill = 0; }
}

typedef struct { uint32_t l, u, m; int32_t scale3; } arith_t;
static void arith_step(arith_t* a, const stats_t* s, uint32_t x, bitw_t* os)     /* arith.cpp:33-104 */
{
    const uint32_t msb_shift = a->m - 1, smsb_shift = a->m - 2, clear = (1u << msb_shift) - 1;
    uint64_t range = (uint64_t)a->u - a->l + 1;
    uint32_t lo = 0, hi, msbL, msbU, e12, e3;
    for (uint32_t i = 0; i < x; ++i) lo += s->counts[i];
    hi = lo + s->counts[x];
    a->u = a->l + (uint32_t)((range * hi) / s->n) - 1;
    a->l = a->l + (uint32_t)((range * lo) / s->n);
    for (;;) {
        msbL = a->l >> msb_shift; msbU = a->u >> msb_shift;
        e12 = msbL == msbU;
        e3 = !e12 && (a->l >> smsb_shift) == 1 && (a->u >> smsb_shift) == 2;
        if (!e12 && !e3) break;
        if (e12) {
            bw_bit(os, msbL);
            a->l = (a->l & clear) << 1;
            a->u = ((a->u & clear) << 1) + 1;
            while (a->scale3 > 0) { bw_bit(os, !msbL); a->scale3--; }
        } else {
            a->scale3++;
            a->u = (((a->u << 1) & clear) | (1u << msb_shift)) + 1;
            a->l = (a->l << 1) & clear;
        }
    }
}
static void stats_update(stats_t* s, uint32_t x, uint32_t r)                    /* qv_stream.cpp:19-35 */
{
    s->counts[x] += s->step; s->n += s->step;
    if (s->n > r) {
        s->n = 0;
        for (uint32_t i = 0; i < s->card; ++i) if (s->counts[i]) { s->counts[i] >>= 1; s->counts[i] += 1; s->n += s->counts[i]; }
    }
}

typedef struct { uint32_t columns; column_t* col; well_t seed; size_t parsed; } book_t;

static void book_free(book_t* b)
{
    if (!b->col) return;
    for (uint32_t c = 0; c < b->columns; ++c) {
        if (b->col[c].st) for (uint32_t j = 0; j < 2 * b->col[c].in.size; ++j) free(b->col[c].st[j].counts);
        free(b->col[c].st); free(b->col[c].q); free(b->col[c].qratio);
    }
    free(b->col); b->col = NULL;
}

static int column_alloc(column_t* c, const alpha_t* in)
{
    c->in = *in;
    c->qratio = (uint8_t*)calloc(in->size, 1);
    c->q = (quant_t*)calloc(2 * in->size, sizeof(quant_t));
    c->st = (stats_t*)calloc(2 * in->size, sizeof(stats_t));
    return (c->qratio && c->q && c->st) ? 0 : -1;
}

static int book_parse(book_t* b, const uint8_t* p, size_t n)
{
    size_t pos = 0; alpha_t uniques, zero;
#define NEED(k) do { if (pos + (k) > n) return -1; } while (0)
    memset(b, 0, sizeof *b);
    NEED(132);
    memcpy(b->seed.state, p, 128); pos = 128;
    memcpy(&b->columns, p + pos, 4); pos += 4;
    if (b->columns == 0 || b->columns >= 255) return -1;
    b->col = (column_t*)calloc(b->columns, sizeof(column_t));
    if (!b->col) return -1;
    zero.size = 1; zero.sym[0] = 0; alpha_index(&zero);                          /* alloc_alphabet(1) */
    if (column_alloc(&b->col[0], &zero)) return -1;
    NEED(3 + 2 * A_SIZE);
    b->col[0].qratio[0] = (uint8_t)(p[pos] - 33); pos += 1;
    if ((((uint32_t)p[pos] << 8) | p[pos + 1]) != A_SIZE) return -1;
    pos += 2;
    if (quant_load(&b->col[0].q[0], p + pos)) return -1;
    pos += A_SIZE;
    if (quant_load(&b->col[0].q[1], p + pos)) return -1;
    pos += A_SIZE;
    if (alpha_union(&b->col[0].q[0].out, &b->col[0].q[1].out, &uniques)) return -1;
    for (uint32_t c = 1; c < b->columns; ++c) {
        uint32_t size = uniques.size, part;
        if (column_alloc(&b->col[c], &uniques)) return -1;
        uniques.size = 0; alpha_index(&uniques);                                 /* alloc_alphabet(0) */
        NEED(2); part = ((uint32_t)p[pos] << 8) | p[pos + 1]; pos += 2;
        NEED(part);
        if (part < size) return -1;
        for (uint32_t i = 0; i < size; ++i) b->col[c].qratio[i] = (uint8_t)(p[pos + i] - 33);
        pos += part;
        for (uint32_t hl = 0; hl < 2; ++hl)
            for (uint32_t i = 0; i < size; ++i) {
                quant_t* z = &b->col[c].q[2 * i + hl];
                NEED(A_SIZE);
                if (quant_load(z, p + pos)) return -1;
                pos += A_SIZE;
                if (alpha_union(&uniques, &z->out, &uniques)) return -1;
            }
    }
    /* initialize_stream_stats: uniform counts, step 8 */
    for (uint32_t c = 0; c < b->columns; ++c)
        for (uint32_t j = 0; j < 2 * b->col[c].in.size; ++j) {
            stats_t* s = &b->col[c].st[j];
            s->card = b->col[c].q[j].out.size; s->step = 8; s->n = s->card;
            if (s->card == 0) return -1;
            s->counts = (uint32_t*)malloc(4 * s->card);
            if (!s->counts) return -1;
            for (uint32_t k = 0; k < s->card; ++k) s->counts[k] = 1;
        }
    b->parsed = pos;
    return 0;
#undef NEED
}

/* number of footer bytes the quality section occupies (WELL state + max_read_length + codebook), or -1 */
long fso_qvz_footer_size(const uint8_t* footer, size_t footer_bytes)
{
    book_t b; long r;
    r = book_parse(&b, footer, footer_bytes) ? -1 : (long)b.parsed;
    book_free(&b);
    return r;
}

/* One block's quality stream.  quals: quality values (offset removed, < 72) of the reads one after another in
 * coding order; lens[n_reads].  Returns the stream size (which may exceed cap: truncated) or -1 on malformed input. */
long fso_qvz_encode(const uint8_t* footer, size_t footer_bytes, const uint8_t* quals, const uint32_t* lens, uint32_t n_reads,
                    uint8_t* out, size_t cap)
{
    book_t b; well_t well; arith_t a; bitw_t os; long ret = -1;
    if (book_parse(&b, footer, footer_bytes)) { book_free(&b); return -1; }
    memset(&well, 0, sizeof well); memcpy(well.state, b.seed.state, sizeof well.state);   /* ResetWellRng */
    a.m = 22; a.l = 0; a.u = (1u << a.m) - 1; a.scale3 = 0;                                /* initialize_arithmetic_encoder */
    os.out = out; os.cap = cap; os.pos = 0; os.cur = 0; os.fill = 0;
    for (uint32_t r = 0; r < n_reads; ++r) {
        uint32_t prev = 0;
        if (lens[r] > b.columns) goto done;
        for (uint32_t i = 0; i < lens[r]; ++i) {
            column_t* c = &b.col[i];
            uint32_t qv = quals[i], idx, qi, hat, state;
            if (qv >= A_SIZE || prev >= INDEX_SLOTS) goto done;
            idx = c->in.index[prev];                                             /* choose_quantizer */
            if (idx == NOT_FOUND) goto done;
            qi = well_bits(&well, 7) >= c->qratio[idx] ? 2 * idx + 1 : 2 * idx;
            hat = c->q[qi].q[qv];
            if (hat >= INDEX_SLOTS) goto done;
            state = c->q[qi].out.index[hat];
            if (state == NOT_FOUND) goto done;
            arith_step(&a, &c->st[qi], state, &os);                              /* compress_qv */
            stats_update(&c->st[qi], state, 1u << (a.m - 3));
            prev = hat;
        }
        quals += lens[r];
    }
    {                                                                            /* encoder_last_step */
        uint32_t msbL = a.l >> (a.m - 1);
        bw_bit(&os, msbL);
        while (a.scale3 > 0) { bw_bit(&os, !msbL); a.scale3--; }
        for (int k = (int)a.m - 2; k >= 0; --k) bw_bit(&os, (a.l >> k) & 1u);
        while (os.fill) bw_bit(&os, 0);
    }
    ret = (long)os.pos;
done:
    book_free(&b);
    return ret;
}
