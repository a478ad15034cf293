// TEST-ONLY stand-in for fastore_amd/csrc/engine.hip: runs the same coder cores (ppmd_core.h,
// rc_core.h) on the host, one "lane", so that archive parity of the whole host pipeline can be
// checked in a container without a GPU.  Built into build/libfastore_emu.so by `make emu`; the
// product library (libfastore_amd.so) never contains or loads this file.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <algorithm>
#include <map>
#include <vector>
#include <atomic>
#include <mutex>
#include <dlfcn.h>
#include "../../fastore_amd/csrc/engine.h"
#include "../../fastore_amd/csrc/ppmd_core.h"
#include "../../fastore_amd/csrc/rc_core.h"
#include "../../fastore_amd/csrc/qvz_core.h"
#include "../../fastore_amd/csrc/emit_core.h"

using namespace fsdev;
namespace fsengine {

int device_count() { return 1; }
int device_create(Device** out, int, uint32_t, char*, size_t)
{ Device* d = new Device(); memset(d, 0, sizeof *d); snprintf(d->name, sizeof d->name, "host-emulation"); d->nWaves = 1; *out = d; return 0; }
int lane_create(Device*, Device** out, char* e, size_t n) { return device_create(out, 0, 0, e, n); }
void device_destroy(Device* d) { if (d) free(d->hStage); delete d; }
int lanes_equalize(Device* const*, size_t) { return 0; }
void set_pageable_staging(bool) {}
bool pageable_staging() { return false; }
int lane_debug(Device*, char* out, size_t outLen) { if (out && outLen) snprintf(out, outLen, "host emulation"); return 0; }

void staging_release(Device* d) { free(d->hStage); d->hStage = nullptr; d->capStage = 0; }
uint8_t* staging_buffer(Device* d, size_t bytes)
{ if (bytes > d->capStage) { free(d->hStage); d->hStage = (uint8_t*)malloc(bytes + bytes / 4 + 4096); d->capStage = bytes + bytes / 4 + 4096; } return d->hStage; }

static void runItems(const uint8_t* input, std::vector<StreamItem>& items, std::vector<uint8_t>& scratch, std::vector<uint32_t>& sizes, BatchTiming* t)
{
    uint64_t sc = 0;
    for (auto& it : items) { it.out_off = sc; sc += ((uint64_t)it.out_cap + 15u) & ~15ull; }
    scratch.assign(sc + 16, 0); sizes.assign(items.size(), 0);
    std::atomic<size_t> next(0);
    const unsigned nt = std::max(1u, std::thread::hardware_concurrency());
    std::vector<std::thread> pool;
    for (unsigned k = 0; k < nt; ++k) pool.emplace_back([&]() {
        uint8_t* arena = (uint8_t*)aligned_alloc(64, (32ull << 20) + 65536);
        fsppmd::Shared* sh = new fsppmd::Shared;
        for (;;) {
            const size_t i = next.fetch_add(1); if (i >= items.size()) break;
            const StreamItem& it = items[i];
            if (it.kind == KIND_PPMD) { if (it.in_len) sizes[i] = fsppmd::encode_member(arena, sh, input + it.in_off, it.in_len, scratch.data() + it.out_off, it.out_cap, nullptr); }
            else if (it.kind == KIND_QVZ) sizes[i] = fsqvz::encode_stream(arena, input + it.aux_off, input + it.in_off, it.in_len, scratch.data() + it.out_off, it.out_cap);
            else sizes[i] = fsrc::encode_model(it.kind - KIND_RC_BASE, arena, input + it.in_off, it.in_len, scratch.data() + it.out_off, it.out_cap);
        }
        delete sh; free(arena);
    });
    for (auto& th : pool) th.join();
    if (t) { t->launches++; t->items += items.size(); for (auto& it : items) { if (it.kind == KIND_PPMD) t->ppmd_symbols += it.in_len; else t->rc_symbols += it.in_len; } }
}

int encode_streams_raw(Device*, const uint8_t* input, size_t, std::vector<StreamItem>& items, std::vector<uint8_t>& raw, std::vector<uint32_t>& sizes, BatchTiming* t)
{ runItems(input, items, raw, sizes, t); return 0; }

static void putBe(uint8_t*& h, uint64_t v, int n) { for (int i = 0; i < n; ++i) *h++ = (uint8_t)(v >> (8 * (n - 1 - i))); }

// what fs_gather_quality does, one string after the other (the staging buffer has room behind the uploaded bytes)
static void gatherQuality(uint8_t* buf, size_t inputBytes, const GatherPlan& g)
{
    if (g.bits != 6u) {                                            // 8-bin / binary: (symbol, context) pairs, positions under 'N' left out
        const QuaPairString* qs = (const QuaPairString*)(buf + g.desc_off);
        uint8_t* out = buf + ((inputBytes + 15u) & ~(size_t)15u);
        const uint8_t* nl = buf + g.n_list_off;
        for (uint32_t i = 0; i < g.n_strings; ++i) {
            uint32_t at = qs[i].dst_off;
            for (uint32_t j = 0; j < qs[i].len; ++j) {
                const uint32_t ii = qs[i].reverse ? qs[i].len - 1u - j : j;
                bool isN = false;
                for (uint32_t t = 0; t < qs[i].n_count; ++t) isN = isN || nl[qs[i].n_off + t] == ii;
                if (isN) continue;
                const uint64_t bit = qs[i].src_bit + (uint64_t)g.bits * ii;
                const uint32_t w = ((uint32_t)buf[bit >> 3] << 8) | buf[(bit >> 3) + 1];
                uint32_t sym = (w >> (16u - g.bits - (uint32_t)(bit & 7u))) & ((1u << g.bits) - 1u);
                if (g.bits == 1u) sym = g.sym_of_bit[sym];
                out[2 * at] = (uint8_t)sym; out[2 * at + 1] = (uint8_t)(j * (g.bits == 3u ? 8u : 2u) / qs[i].len); ++at;
            }
        }
        return;
    }
    if (g.qvz) {                                                   // --lossy: (context | state << 24) words, what fs_gather_quality_qvz writes
        const QuaQvzString* qs = (const QuaQvzString*)(buf + g.desc_off);
        uint8_t* out = buf + ((inputBytes + 15u) & ~(size_t)15u);
        for (uint32_t i = 0; i < g.n_strings; ++i) {
            const QuaQvzString& d = qs[i];
            const uint8_t* model = buf + 16ull * d.model16;
            QvzSymHeader h; memcpy(&h, model, sizeof h);
            const uint32_t* colCtxBase = (const uint32_t*)(model + h.col_ctx_base_off); const uint16_t* colIndex = (const uint16_t*)(model + h.col_index_off);
            const uint8_t* qratio = model + h.qratio_off; const uint8_t* quant = model + h.quant_off; const uint8_t* stateOf = model + h.state_of_off;
            const uint32_t* well = (const uint32_t*)(model + h.well_off);
            uint32_t prev = 0; bool bad = d.len > h.columns;
            for (uint32_t j = 0; j < d.len; ++j) {
                uint32_t w = 0xFFFFFFFFu;
                if (!bad) {
                    const uint64_t bit = d.src_bit + 6ull * (d.reverse ? d.len - 1u - j : j);
                    const uint32_t two = ((uint32_t)buf[bit >> 3] << 8) | buf[(bit >> 3) + 1];
                    const uint32_t qv = (two >> (10u - (uint32_t)(bit & 7u))) & 63u;
                    const uint32_t idx = prev < 82u ? colIndex[j * 82u + prev] : 0xFFFFu;
                    const uint32_t dn = d.draw0 + j;
                    if (idx == 0xFFFFu || (dn >> 2) >= h.well_words) bad = true;
                    else {
                        const uint32_t pair = colCtxBase[j] / 2u + idx;
                        const uint32_t ctx = 2u * pair + (((well[dn >> 2] >> (7u * (dn & 3u))) & 127u) >= qratio[pair] ? 1u : 0u);
                        if (ctx >= h.n_ctx || stateOf[ctx * 72u + qv] == 0xFFu) bad = true;
                        else { w = ctx | ((uint32_t)stateOf[ctx * 72u + qv] << 24); prev = quant[ctx * 72u + qv]; }
                    }
                }
                memcpy(out + d.dst_off + 4ull * j, &w, 4);
            }
        }
        return;
    }
    const QuaString* qs = (const QuaString*)(buf + g.desc_off);
    uint8_t* out = buf + ((inputBytes + 15u) & ~(size_t)15u);
    for (uint32_t i = 0; i < g.n_strings; ++i)
        for (uint32_t j = 0; j < qs[i].len; ++j) {
            const uint64_t bit = qs[i].src_bit + 6ull * (qs[i].reverse ? qs[i].len - 1u - j : j);
            const uint32_t w = ((uint32_t)buf[bit >> 3] << 8) | buf[(bit >> 3) + 1];
            out[qs[i].dst_off + j] = (uint8_t)((w >> (10u - (uint32_t)(bit & 7u))) & 63u);
        }
}

// what matcher.hip computes, as a plain scalar loop over a window kept newest first (test-only stand-in)
struct MatchLane { int unused; };
int match_lane_create(Device*, MatchLane** out, bool) { *out = new MatchLane(); return 0; }
void match_lane_destroy(MatchLane* m) { delete m; }
int match_lane_reserve(Device*, MatchLane*, size_t, size_t, size_t, size_t) { return 0; }
static std::atomic<uint64_t> g_unpackWords{0}, g_unpackDiff{0};
void unpack_check(bool on) { if (on) { g_unpackWords = 0; g_unpackDiff = 0; } }
void unpack_check_counts(uint64_t* words, uint64_t* differing) { *words = g_unpackWords.load(); *differing = g_unpackDiff.load(); }
int match_reads(Device* dev, MatchLane*, const uint8_t* seqIn, size_t seqBytes, const PackedDna* packed, const MatchRead* reads, size_t nReads, const MatchCall* calls, size_t nCalls,
                const uint32_t* warm, size_t, const MatchParams& par, MatchRow* rows, double*)
{
    // packed bases: what fs_unpack_planes reads, base by base (the stand-in searches on THESE bases, and holds them against the
    // host's unpacked ones: a wrong descriptor shows as a differing base here and as different rows in the parity tests)
    std::vector<uint8_t> own;
    const uint8_t* seq = seqIn;
    if (packed) {
        own.assign(seqBytes, 0);
        for (size_t i = 0; i < nReads; ++i) {
            const PackedRead& q = packed->reads[i]; const MatchRead& rd = reads[i];
            const bool plain = (q.info & PACKED_PLAIN) != 0u, hasSig = (q.info & PACKED_HAS_SIG) != 0u;
            const uint32_t sigId = q.info & ((1u << PACKED_SIG_BITS) - 1u), sigPos = (q.info >> PACKED_SIG_BITS) & 0xFFu, hole = hasSig ? packed->sig_len : 0u, bits = plain ? 2u : 3u;
            for (uint32_t pos = 0; pos < rd.len; ++pos) {
                uint32_t code;
                if (hasSig && pos >= sigPos && pos < sigPos + hole) code = (sigId >> (2u * (hole - 1u - (pos - sigPos)))) & 3u;
                else {
                    const uint32_t j = pos < sigPos || !hasSig ? pos : pos - hole;
                    const uint64_t at = (uint64_t)q.bit_off + (uint64_t)bits * j;
                    if (at + bits > 8ull * packed->bytes) { snprintf(dev->err, sizeof dev->err, "device matcher: read %zu outside the packed bases", i); return -1; }
                    code = 0;
                    for (uint32_t b = 0; b < bits; ++b) { const uint64_t x = at + b; code = (code << 1) | ((packed->dna[x >> 3] >> (7u - (uint32_t)(x & 7u))) & 1u); }
                }
                const uint8_t c = code < 5u ? packed->symbol_order[code] : 0;
                own[rd.seq_off + pos] = c;
                ++g_unpackWords; if (c != seqIn[rd.seq_off + pos]) ++g_unpackDiff;
            }
        }
        seq = own.data();
    }
    const uint32_t cap = par.window - 1u;
    for (size_t c = 0; c < nCalls; ++c) {
        const MatchCall& call = calls[c];
        std::vector<uint32_t> win;                                     // newest first
        if (call.aux >= 0) win.push_back((uint32_t)call.aux);
        for (uint32_t k = 0; k < call.warm_count; ++k) win.insert(win.begin(), warm[call.warm_first + k]);     // listed oldest first
        for (uint32_t i = 0; i < call.count; ++i) {
            const uint32_t r = call.first + i; const MatchRead& rd = reads[r];
            const int32_t thr = par.encode_threshold ? par.encode_threshold : (int32_t)(rd.len / 2u);
            int32_t best = thr + 1, bestShift = 0; int64_t bestSlot = -1; bool bestNoMism = false; uint32_t bestLen = 0;
            auto price = [&](const uint8_t* e, uint32_t eLen, int32_t eMin, int64_t slot) {
                const int32_t shift = eMin - (int32_t)rd.min_pos, ashift = shift < 0 ? -shift : shift;
                if (ashift > 127) return;
                const uint32_t recOff = shift < 0 ? (uint32_t)ashift : 0u, lzOff = shift > 0 ? (uint32_t)ashift : 0u;
                const uint32_t n = std::min(rd.len - recOff, eLen - lzOff);
                int32_t mism = 0;
                for (uint32_t k = 0; k < n; ++k) mism += seq[rd.seq_off + recOff + k] != e[lzOff + k];
                const int32_t cc = ashift * par.shift_cost + mism * par.mismatch_cost;
                if (cc < best) { best = cc; bestShift = shift; bestSlot = slot; bestNoMism = mism == 0; bestLen = eLen; }
            };
            for (size_t j = 0; j < win.size(); ++j) price(seq + reads[win[j]].seq_off, reads[win[j]].len, reads[win[j]].min_pos, (int64_t)j);
            uint8_t dummy[256]; memset(dummy, 'N', sizeof dummy);
            if (win.size() < cap) price(dummy, 256u, 0, (int64_t)win.size());
            MatchRow row{-1, (int16_t)(thr + 1), 0, 0, 0, 0, 0};
            bool identical = false;
            if (bestSlot >= 0) {
                const bool isDummy = (size_t)bestSlot >= win.size();
                row.match = isDummy ? -2 : (int32_t)win[(size_t)bestSlot]; row.cost = (int16_t)best; row.shift = (int16_t)bestShift;
                row.no_mismatches = bestNoMism; row.dummy = isDummy;
                identical = best == 0 && bestLen == rd.len && !isDummy && row.match != call.aux;
                row.identical = identical;
            }
            rows[r] = row;
            if (!identical) { win.insert(win.begin(), r); if (win.size() > cap) win.pop_back(); }
        }
    }
    return 0;
}

// what fs_match_mates computes, as a plain loop over a history kept newest first (test-only stand-in; the rules are those of
// LzCompressorPE::CompressPair, restated from the kernel's description, not from the product's host search)
int match_mates(Device*, MatchLane*, const uint8_t* seq, size_t, const MatePair* pairs, size_t nPairs, const uint32_t* validBits, size_t, const MateParams& par, MateRow* rows, double*);
int match_mates_batch(Device* dev, MatchLane* m, const MateBatchJob* jobs, size_t nJobs, const uint32_t* validBits, size_t validWords, const MateParams& par, double* kms)
{
    for (size_t j = 0; j < nJobs; ++j)
        if (jobs[j].nPairs && match_mates(dev, m, jobs[j].seq, jobs[j].seqBytes, jobs[j].pairs, jobs[j].nPairs, validBits, validWords, par, jobs[j].rows, kms) != 0) return -1;
    return 0;
}
int match_mates(Device*, MatchLane*, const uint8_t* seq, size_t, const MatePair* pairs, size_t nPairs, const uint32_t* validBits, size_t, const MateParams& par, MateRow* rows, double*)
{
    if (const char* lib = getenv("FS_EMU_SIMT_MATES")) {              // the kernel's own body on the lock-step wave emulation (tests/emu/mates_simt.cpp) instead of the plain loop below
        typedef int (*Fn)(const uint8_t*, const MatePair*, uint32_t, const uint32_t*, const MateParams*, MateRow*);
        static const Fn fn = [lib]() { void* h = dlopen(lib, RTLD_NOW | RTLD_LOCAL); return h ? (Fn)dlsym(h, "simt_match_mates") : (Fn) nullptr; }();      // (the emulation keeps a set of fibers per calling thread)
        if (!fn) return -1;
        return fn(seq, pairs, (uint32_t)nPairs, validBits, &par, rows);
    }
    struct Entry { uint32_t sig[4]; uint32_t pos[4]; uint32_t off, len; int32_t pair; bool live; };
    std::vector<Entry> hist;                                           // newest first; only entries that went to the front
    uint8_t idx[128]; memset(idx, 255, sizeof idx);
    for (int k = 0; k < 5; ++k) idx[par.symbol_order[k] & 127] = (uint8_t)k;
    const uint32_t L = par.sig_len;
    for (size_t p = 0; p < nPairs; ++p) {
        const MatePair& pr = pairs[p];
        const uint8_t* m8 = seq + pr.mate_off; const int32_t plen = pr.mate_len, half = plen / 2;
        if (hist.size() >= par.window) hist.pop_back();                // the oldest leaves (an entry that went to the back never got in)
        const int32_t end1 = plen - (int32_t)L - ((int32_t)par.skip_zone + half - ((int32_t)L - 1)), end2 = plen - (int32_t)L - (int32_t)par.skip_zone;
        std::map<uint32_t, uint32_t> set1, set2;                       // signature -> first position
        auto sigAt = [&](int32_t t, uint32_t& m) { m = 0; for (uint32_t k = 0; k < L; ++k) { const uint32_t c = idx[m8[t + k] & 127]; if (c > 3) return false; m = (m << 2) | c; } return ((validBits[m >> 5] >> (m & 31u)) & 1u) != 0u; };
        for (int32_t t = 0; t < end1; ++t) { uint32_t m; if (sigAt(t, m) && !set1.count(m)) set1[m] = (uint32_t)t; }
        for (int32_t t = half; t < end2; ++t) { uint32_t m; if (sigAt(t, m) && !set1.count(m) && !set2.count(m)) set2[m] = (uint32_t)t; }
        std::map<uint32_t, uint32_t> all = set1; all.insert(set2.begin(), set2.end());
        int32_t best = 255, bestShift = 0; int64_t bestAt = -1; bool bestNoMism = false;
        for (const auto& sp : all)                                      // signatures ascending
            for (size_t h = hist.size(); h-- > 0;) {                   // entries oldest first
                const Entry& e = hist[h];
                bool lists = false; for (int k = 0; k < 4; ++k) lists = lists || (e.sig[k] != 0 && e.sig[k] == sp.first);
                if (!lists) continue;
                for (int k = 0; k < 4; ++k) {
                    const int32_t shift = (int32_t)e.pos[k] - (int32_t)sp.second, ashift = shift < 0 ? -shift : shift;
                    if (ashift > 127) continue;
                    const uint32_t recOff = shift < 0 ? (uint32_t)ashift : 0u, lzOff = shift > 0 ? (uint32_t)ashift : 0u;
                    const uint32_t n = std::min<uint32_t>((uint32_t)plen - recOff, e.len - lzOff);
                    int32_t mism = 0;
                    for (uint32_t i = 0; i < n; ++i) mism += m8[recOff + i] != seq[e.off + lzOff + i];
                    const int32_t cc = ashift * par.shift_cost + mism * par.mismatch_cost;
                    if (cc < best) { best = cc; bestShift = shift; bestAt = (int64_t)h; bestNoMism = mism == 0; }
                }
            }
        MateRow row; memset(&row, 0, sizeof row);
        row.cost = (int16_t)best; row.shift = (int16_t)bestShift; row.no_mismatches = bestNoMism; row.match = -1;
        const bool matched = best <= pr.threshold;
        if (matched) { row.match = hist[(size_t)bestAt].pair; row.prev_id = (uint16_t)bestAt; }
        rows[p] = row;
        const bool identical = matched && bestNoMism && best == 0;
        if (!identical) {
            Entry e; memset(&e, 0, sizeof e);
            e.off = pr.mate_off; e.len = (uint32_t)plen; e.pair = (int32_t)p; e.live = true;
            const std::map<uint32_t, uint32_t>& first = set1.size() > set2.size() ? set2 : set1; const std::map<uint32_t, uint32_t>& second = set1.size() > set2.size() ? set1 : set2;
            uint32_t n = 0;
            for (auto it = first.begin(); it != first.end() && n < 2u; ++it, ++n) { e.sig[n] = it->first; e.pos[n] = it->second; }
            for (auto it = second.begin(); it != second.end() && n < 4u; ++it, ++n) { e.sig[n] = it->first; e.pos[n] = it->second; }
            hist.insert(hist.begin(), e);
        } else if (hist.size() + 1 >= par.window && !hist.empty()) {
            // the mate sits at the back until the next pair drops it -- IN PLACE of the oldest entry, which the pop in front of this
            // search has already taken away: nothing to undo
        }
    }
    return 0;
}

// what fs_tokenise_ids does, one read id after the other (work = the device's input buffer with room behind the input)
static void tokeniseIds(uint8_t* buf, size_t inputBytes, const IdPlan& p, std::vector<StreamItem>& items)
{
    const IdJob* jobs = (const IdJob*)(buf + p.jobs_off); const IdString* ss = (const IdString*)(buf + p.strings_off);
    uint8_t* out = buf + ((inputBytes + 15u) & ~(size_t)15u);
    for (uint32_t j = 0; j < p.n_jobs; ++j) {
        const IdJob& job = jobs[j];
        const uint8_t* tab = buf + job.table_off;
        const uint32_t nf = *(const uint32_t*)tab; const IdField* F = (const IdField*)(tab + 8);
        uint32_t nt = 0, nv = 0;
        for (uint32_t r = 0; r < job.count; ++r) {
            const IdString& s = ss[job.first + r];
            std::vector<uint8_t> h(s.len);
            for (uint32_t k = 0; k < s.len; ++k) {
                if (k == 0) { h[0] = '@'; continue; }
                const uint64_t bit = s.src_bit + 7ull * (k - 1);
                h[k] = (uint8_t)(((((uint32_t)buf[bit >> 3] << 8) | buf[(bit >> 3) + 1]) >> (9u - (uint32_t)(bit & 7u))) & 127u);
            }
            uint32_t fieldStart = 0, fi = 0;
            for (uint32_t i = 0; i <= s.len; ++i) {
                if (fi >= nf) break;
                const IdField& f = F[fi];
                if (i != s.len && h[i] != f.separator) continue;
                if (!f.is_const) {
                    const uint32_t fl = i - fieldStart;
                    if (!f.is_numeric) {
                        uint32_t id = f.n_values; const uint32_t* vl = (const uint32_t*)(tab + f.values_off);
                        for (uint32_t v = 0; v < f.n_values && id == f.n_values; ++v) if (vl[2 * v + 1] == fl && memcmp(tab + vl[2 * v], h.data() + fieldStart, fl) == 0) id = v;
                        out[job.tok_out + 2ull * nt] = (uint8_t)id; out[job.tok_out + 2ull * nt + 1] = (uint8_t)fi; ++nt;
                    } else {
                        uint64_t v = 0;
                        for (uint32_t k = 0; k < fl; ++k) { const uint8_t c = h[fieldStart + k]; if (c < '0' || c > '9') break; v = v * 10 + (c - '0'); }
                        const int64_t diff = (int64_t)(v - f.min_value); uint32_t ctx = fi << 2;
                        for (int32_t q = (int32_t)f.plog; q >= 0; --q) { out[job.val_out + 2ull * nv] = (uint8_t)((diff >> (8 * q)) & 0xFF); out[job.val_out + 2ull * nv + 1] = (uint8_t)ctx; ++ctx; ++nv; }
                    }
                }
                fieldStart = i + 1; ++fi;
            }
        }
        items[job.tok_item].in_len = nt; items[job.val_item].in_len = nv;
    }
}

int tokenise_ids_raw(Device*, const uint8_t* input, size_t inputBytes, const IdPlan& plan, std::vector<std::vector<uint8_t>>& tok, std::vector<std::vector<uint8_t>>& val)
{
    std::vector<uint8_t> work(((inputBytes + 15u) & ~(size_t)15u) + plan.out_bytes + 64);
    memcpy(work.data(), input, inputBytes);
    std::vector<StreamItem> items(2 * (size_t)plan.n_jobs); memset(items.data(), 0, items.size() * sizeof(StreamItem));
    tokeniseIds(work.data(), inputBytes, plan, items);
    const IdJob* jb = (const IdJob*)(input + plan.jobs_off);
    const uint8_t* out = work.data() + ((inputBytes + 15u) & ~(size_t)15u);
    tok.assign(plan.n_jobs, {}); val.assign(plan.n_jobs, {});
    for (uint32_t j = 0; j < plan.n_jobs; ++j) {
        tok[j].assign(out + jb[j].tok_out, out + jb[j].tok_out + 2ull * items[2 * j].in_len);
        val[j].assign(out + jb[j].val_out, out + jb[j].val_out + 2ull * items[2 * j + 1].in_len);
    }
    return 0;
}

int gather_quality_raw(Device*, const uint8_t* input, size_t inputBytes, const GatherPlan& plan, std::vector<uint8_t>& out, BatchTiming* t)
{
    std::vector<uint8_t> work(((inputBytes + 15u) & ~(size_t)15u) + plan.out_bytes + 64);
    memcpy(work.data(), input, inputBytes);
    gatherQuality(work.data(), inputBytes, plan);
    out.assign(work.begin() + ((inputBytes + 15u) & ~(size_t)15u), work.begin() + ((inputBytes + 15u) & ~(size_t)15u) + plan.out_bytes);
    if (t) t->gather_symbols += plan.symbols;
    return 0;
}

// what fs_emit_count / fs_emit_scan / fs_emit_write / fs_rle_binary / fs_rle0 do, one op after the other (emit_core.h is the kernels' own source)
static void emitStreams(uint8_t* buf, const EmitPlan& plan, uint64_t emitBase, std::vector<StreamItem>& items)
{
    const EmitJob* jobs = (const EmitJob*)(buf + plan.jobs_off); const EmitOp* ops = (const EmitOp*)(buf + plan.ops_off);
    const uint32_t* ids = (const uint32_t*)(buf + plan.ids_off);
    uint8_t* out = buf + emitBase;
    for (uint32_t j = 0; j < plan.n_jobs; ++j) {
        const EmitJob& job = jobs[j];
        uint32_t at[ECH_COUNT] = {0};
        // FS_EMU_SIMT_EMIT=<build/libsimt_emu.so>: the kernels' own body -- a wavefront per op, the match bits as packed ballots (emit_wave.h) -- on the
        // lock-step wave emulation (tests/emu/emit_simt.cpp) instead of the serial form below
        typedef int (*SimtEmit)(const uint8_t*, const EmitJob*, const EmitOp*, uint8_t*, uint32_t*);
        static const SimtEmit simtEmit = []() -> SimtEmit { const char* lib = getenv("FS_EMU_SIMT_EMIT"); if (!lib) return nullptr; void* h = dlopen(lib, RTLD_NOW | RTLD_LOCAL); return h ? (SimtEmit)dlsym(h, "simt_emit_job") : (SimtEmit) nullptr; }();
        if (getenv("FS_EMU_SIMT_EMIT")) {
            if (!simtEmit || simtEmit(buf, &job, ops, out, at) != 0) { fprintf(stderr, "FS_EMU_SIMT_EMIT: the emulated emission kernels could not run\n"); abort(); }
        } else
        for (uint32_t k = 0; k < job.n_ops; ++k) {
            const EmitOp& op = ops[job.first_op + k];
            const uint32_t chL = fsemit::channel_l(op), chB = fsemit::channel_b(op);
            fsemit::Sink s; uint8_t dummy[4];
            s.outL = chL < ECH_COUNT ? out + job.out_off[chL] + (uint64_t)fsemit::unit_l(chL) * at[chL] : dummy;
            s.outB = chB < ECH_COUNT ? (fsemit::is_bit_channel(chB) ? out + job.raw_off[chB] + at[chB] : out + job.out_off[chB] + 2ull * at[chB]) : dummy;
            fsemit::emit_op(op, job, buf + job.seq_off, buf + job.contig_off, s);
            if (chL < ECH_COUNT) at[chL] += s.nL;
            if (chB < ECH_COUNT) at[chB] += s.nB;
        }
        for (uint32_t c = 0; c < ECH_COUNT; ++c) {
            if (job.item[c] == 0xFFFFFFFFu) continue;
            items[job.item[c]].in_len = fsemit::is_bit_channel(c) ? fsemit::rle_binary_serial(out + job.raw_off[c], at[c], out + job.out_off[c]) : at[c];
        }
        if (job.item[ECH_COUNT] != 0xFFFFFFFFu) items[job.item[ECH_COUNT]].in_len = fsemit::rle0_serial(ids + job.first_id, job.n_ids, out + job.out_off[ECH_COUNT]);
    }
}

int emit_streams_raw(Device* dev, const uint8_t* input, size_t inputBytes, const EmitPlan& plan, std::vector<StreamItem>& items, std::vector<std::vector<std::vector<uint8_t>>>& streams)
{
    uint32_t badJob = 0;
    if (const char* why = fsemit::plan_error(input, inputBytes, plan, (uint32_t)items.size(), badJob)) { snprintf(dev->err, sizeof dev->err, "emission job %u: %s", badJob, why); return -1; }
    const uint64_t emitBase = ((uint64_t)inputBytes + 15u) & ~15ull;
    std::vector<uint8_t> work(emitBase + plan.out_bytes + 64);
    memcpy(work.data(), input, inputBytes);
    emitStreams(work.data(), plan, emitBase, items);
    const EmitJob* jobs = (const EmitJob*)(input + plan.jobs_off);
    streams.assign(plan.n_jobs, std::vector<std::vector<uint8_t>>(ECH_COUNT + 1));
    for (uint32_t j = 0; j < plan.n_jobs; ++j)
        for (uint32_t c = 0; c <= ECH_COUNT; ++c) {
            if (jobs[j].item[c] == 0xFFFFFFFFu) continue;
            const uint32_t len = items[jobs[j].item[c]].in_len;
            const uint64_t n = (c == ECH_COUNT || fsemit::is_bit_channel(c) || c == ECH_HARD || c == ECH_HARD_PE) ? (uint64_t)len : 2ull * len;
            streams[j][c].assign(work.begin() + (ptrdiff_t)(emitBase + jobs[j].out_off[c]), work.begin() + (ptrdiff_t)(emitBase + jobs[j].out_off[c] + n));
        }
    return 0;
}

int encode_batch(Device* dev, const uint8_t* input, size_t inputBytes, std::vector<StreamItem>& items, std::vector<BlockPlan>& plans,
                 std::vector<uint8_t>& blocks, std::vector<uint64_t>& blockSizes, BatchTiming* t, const GatherPlan* gather, const IdPlan* ids, const EmitPlan* emit)
{
    std::vector<uint8_t> scratch; std::vector<uint32_t> sizes;
    std::vector<uint8_t> work;                          // the device's input buffer: uploaded bytes + gather region
    if ((gather && gather->n_strings) || (ids && ids->n_jobs) || (emit && emit->n_jobs)) {
        const uint64_t emitBase = ((inputBytes + 15u) & ~(size_t)15u) + (gather ? gather->out_bytes : 0) + (ids ? ids->out_bytes : 0);
        work.resize(emitBase + (emit ? emit->out_bytes : 0) + 64);
        memcpy(work.data(), input, inputBytes);
        if (gather && gather->n_strings) gatherQuality(work.data(), inputBytes, *gather);
        if (ids && ids->n_jobs) tokeniseIds(work.data(), inputBytes, *ids, items);
        if (emit && emit->n_jobs) {
            uint32_t badJob = 0;
            if (const char* why = fsemit::plan_error(input, inputBytes, *emit, (uint32_t)items.size(), badJob)) { snprintf(dev->err, sizeof dev->err, "emission job %u: %s", badJob, why); return -1; }
            emitStreams(work.data(), *emit, emitBase, items);
        }
        input = work.data();
        if (t && gather) { t->gather_symbols += gather->symbols; }
        if (t && ids) t->id_strings += ids->n_strings;
    }
    runItems(input, items, scratch, sizes, t);
    for (size_t i = 0; i < items.size(); ++i) {
        if (sizes[i] == 0xFFFFFFFFu) { snprintf(dev->err, sizeof dev->err, "stream item %zu: symbol or context outside its coder's alphabet (corrupted input)", i); return -2; }
        if (sizes[i] >= items[i].out_cap && items[i].in_len > 0) { snprintf(dev->err, sizeof dev->err, "stream overflow"); return -2; }
    }
    blocks.clear(); blockSizes.assign(plans.size(), 0);
    for (size_t b = 0; b < plans.size(); ++b) {
        BlockPlan& pl = plans[b];
        const uint32_t N = pl.n_streams; const uint64_t headerSize = 42ull + 16ull * N;
        uint64_t sz = headerSize + 1; for (uint32_t s = 0; s < N; ++s) sz += sizes[pl.first_item + s];
        pl.block_off = blocks.size(); blockSizes[b] = sz; blocks.resize(blocks.size() + sz, 0);
        uint8_t* blk = blocks.data() + pl.block_off; uint8_t* h = blk;
        uint64_t pos = headerSize; uint64_t dstOff[MAX_STREAMS];
        for (uint32_t k = 0; k < N; ++k) { const uint32_t s = pl.copy_order[k]; dstOff[s] = pos; pos += sizes[pl.first_item + s]; }
        putBe(h, pl.signature, 4); putBe(h, pl.records, 8); *h++ = pl.min_len; *h++ = pl.max_len; putBe(h, pl.raw_dna_size, 8); putBe(h, pos, 8); putBe(h, 1, 4);
        if (pl.has_headers) putBe(h, pl.raw_id_size, 8);
        for (uint32_t s = 0; s < N; ++s) putBe(h, pl.work_size[s] == ~0ull ? (uint64_t)sizes[pl.first_item + s] : (pl.work_size[s] == ~1ull ? (uint64_t)items[pl.first_item + s].in_len : pl.work_size[s]), 8);
        for (uint32_t s = 0; s < N; ++s) putBe(h, sizes[pl.first_item + s], 8);
        for (uint32_t s = 0; s < N; ++s) memcpy(blk + dstOff[s], scratch.data() + items[pl.first_item + s].out_off, sizes[pl.first_item + s]);
        blk[pos] = 0;
    }
    return 0;
}
}  // namespace fsengine
