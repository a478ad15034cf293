// Design study (host only): how long the runs of "plain hits" are in a PPMd stream -- symbols coded in their first
// context at full order whose found state leads to a real context, so that the model needs no update -- and how many
// positions of a 64-symbol window share a context.  These runs are what the device's windowed hit path covers.
// Build: g++ -O2 -std=c++17 -Ifastore_amd/csrc -o build/ppmd_windows tools/ppmd_windows.cpp
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <random>
#include <vector>
#include <map>
#include <algorithm>
struct SymRec { uint32_t ctx; uint8_t plain, ns, rescaled; };
static std::vector<SymRec> g_syms;
static unsigned g_rescales_before = 0, g_rescale_count = 0;
#define FS_SYMHOOK(firstCtx, lastCtx, rec, coder, succ) do { SymRec r_; r_.ctx = (firstCtx); \
    r_.plain = ((firstCtx) == (lastCtx) && (coder).OrderFall == 0 && (succ) >= (coder).UnitsStart) ? 1 : 0; r_.ns = (uint8_t)(rec).ns(); \
    r_.rescaled = 0; g_syms.push_back(r_); } while (0)
#include "wave.h"
#include "ppmd_core.h"
int main(int argc, char** argv)
{
    std::vector<uint8_t> in;
    if (argc > 2) { FILE* f = fopen(argv[2], "rb"); if (!f) return 1; uint8_t b[65536]; size_t k; while ((k = fread(b, 1, sizeof b, f)) > 0) in.insert(in.end(), b, b + k); fclose(f); }
    else {
        std::mt19937 rng(1); const int n = argc > 1 ? atoi(argv[1]) : 300000; in.resize(n);
        const int steps[8] = {-3, -1, 0, 0, 0, 0, 1, 1}; int cur = 38;
        for (int i = 0; i < n; i++) { if (i % 150 == 0) cur = 38; cur += steps[rng() % 8]; if (cur > 40) cur = 40; if (cur < 2) cur = 2; in[i] = (uint8_t)cur; }
    }
    const size_t n = in.size();
    std::vector<uint8_t> out(n + n / 8 + 1024); uint8_t* arena = (uint8_t*)aligned_alloc(64, (fsppmd::ARENA_BYTES + 63) & ~63ull);
    fsppmd::Shared* sh = new fsppmd::Shared; uint32_t rs = 0;
    const uint32_t sz = fsppmd::encode_member(arena, sh, in.data(), (uint32_t)n, out.data(), (uint32_t)out.size(), &rs);
    printf("n %zu -> %u bytes, %u restarts, %zu hooks\n", n, sz, rs, g_syms.size());
    // greedy windows: at a plain symbol open a window of up to 64 plain symbols with ns in 1..7; anything else is a serial symbol
    uint64_t windows = 0, covered = 0, serial = 0, rounds = 0, hist[8] = {0};
    for (size_t i = 0; i < g_syms.size();) {
        const SymRec& s = g_syms[i];
        if (!(s.plain && s.ns >= 1 && s.ns <= 7)) { ++serial; ++i; continue; }
        size_t L = 0; std::map<uint32_t, int> mult; int mx = 0;
        while (L < 64 && i + L < g_syms.size()) { const SymRec& t = g_syms[i + L]; if (!(t.plain && t.ns >= 1 && t.ns <= 7)) break; mx = std::max(mx, ++mult[t.ctx]); ++L; }
        ++windows; covered += L; rounds += mx; hist[L >= 64 ? 7 : L / 8 > 6 ? 6 : L / 8]++;
        i += L;
    }
    printf("windows %llu covering %llu symbols (%.1f per window, %.2f rounds per window), serial symbols %llu (%.3f %%)\n", (unsigned long long)windows, (unsigned long long)covered,
           (double)covered / windows, (double)rounds / windows, (unsigned long long)serial, 100.0 * serial / g_syms.size());
    printf("window length histogram (0-7, 8-15, ..., 48-63, 64):"); for (int k = 0; k < 8; ++k) printf(" %llu", (unsigned long long)hist[k]); printf("\n");
    return 0;
}
