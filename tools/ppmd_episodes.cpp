// Design study (host only): what the serial symbols of a PPMd stream are made of -- how many leave their first context by an
// escape, at which level they are found, whether the found state leads to a real context, whether a rescale or a new unit is
// due -- i.e. how many would fit a "wide" episode that fetches the suffix chain in one go (PPMd section).
// Build: g++ -O2 -std=c++17 -Ifastore_amd/csrc -o build/ppmd_episodes tools/ppmd_episodes.cpp
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <random>
#include <vector>
#include <map>
#include <string>
#include <algorithm>
static std::map<std::string, unsigned long long> g_cls;
static unsigned long long g_total = 0;
struct HookCtx;
#define FS_SYMHOOK(firstCtx, lastCtx, rec, coder, succ) fs_hook((firstCtx), (lastCtx), (rec), (coder), (succ))
#include "wave.h"
namespace fsppmd { struct Coder; struct Ctx; }
template <class R, class C> static void fs_hook(uint32_t firstCtx, uint32_t lastCtx, const R& rec, const C& m, uint32_t succ);
#include "ppmd_core.h"
template <class R, class C> static void fs_hook(uint32_t firstCtx, uint32_t lastCtx, const R& rec, const C& m, uint32_t succ)
{
    ++g_total;
    // levels walked from the first context to the one the symbol was found in
    uint32_t levels = 0, c = firstCtx;
    while (c != lastCtx && levels < 8) { c = *(const uint32_t*)(m.hb + c + 8); ++levels; }
    const int ofBefore = m.OrderFall - (int)levels;
    const uint32_t ns1 = *(const uint8_t*)(m.hb + firstCtx);
    const bool real = succ >= m.UnitsStart;
    char key[160];
    if (levels == 0 && ofBefore == 0 && real && ns1 >= 1 && ns1 <= 7) { g_cls["plain (window)"]++; return; }
    const char* first = ns1 == 0 ? "bin" : (ns1 <= 7 ? "small" : (ns1 <= 15 ? "mid" : "big"));
    const uint32_t nsF = rec.ns();
    const char* found = nsF == 0 ? "bin" : (nsF <= 15 ? "<=16" : (nsF <= 63 ? "<=64" : "big"));
    snprintf(key, sizeof key, "OF0=%d first=%s levels=%u foundIn=%s succ=%s freq%s", ofBefore, first, levels, found, real ? "ctx" : (succ ? "raw" : "null"), m.fsFreq > 124 ? ">MAX" : "ok");
    g_cls[key]++;
}
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
    printf("n %zu -> %u bytes, %u restarts, %llu symbols\n", n, sz, rs, g_total);
    std::vector<std::pair<unsigned long long, std::string>> v;
    for (auto& kv : g_cls) v.push_back({kv.second, kv.first});
    std::sort(v.rbegin(), v.rend());
    for (auto& e : v) printf("%10llu  %6.3f %%  %s\n", e.first, 100.0 * e.first / g_total, e.second.c_str());
    return 0;
}
