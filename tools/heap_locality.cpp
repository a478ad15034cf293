// Design study (host, lock-step emulation of the 64-lane code): which addresses of the PPMd heap does the walk of a long
// quality stream touch, and how much of that would an address range held in LDS catch?  Contexts are allocated from the top
// of the heap downwards, state lists from UnitsStart upwards, both in order of creation -- and the contexts a predictable
// stream lives in are created early.
//   g++ -O2 -std=c++17 -DFS_SIMT_EMU -DFS_HEAP_STATS -o build/heap_locality tools/heap_locality.cpp tests/emu/simt.cpp
//   build/heap_locality <symbols> [read length] [seed]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../tests/emu/simt.h"
static std::vector<uint32_t> g_ser, g_win;          // per 12-byte unit of the heap
static int g_windowFlag = 0;
static inline void fs_heap_stat_window(int on) { if (simt::lane() == 0) g_windowFlag = on; }
static inline uint32_t fs_heap_stat(uint32_t ix)
{
    const uint32_t u = (ix - 1u) / 12u;
    if (u >= g_ser.size()) return ix;
    if (g_windowFlag) g_win[u]++;                   // a window's accesses are per lane
    else if (simt::lane() == 0) g_ser[u]++;          // the serial walk's are wave-uniform
    return ix;
}
static unsigned long long g_q; static inline void simt_count_quick_rescale() { g_q++; }
#include "../fastore_amd/csrc/ppmd_core.h"

int main(int argc, char** argv)
{
    const size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000;
    const int L = argc > 2 ? atoi(argv[2]) : 150; uint64_t seed = argc > 3 ? strtoull(argv[3], 0, 10) : 1;
    std::vector<uint8_t> in(n + 64, 0), out(2 * n + 4096);
    // the generator's quality model (tools/gen_fastq.cpp): bounded random walk from 38, steps {-3,-1,0,0,0,0,+1,+1}; every other
    // read reversed (a reverse-complemented record's scores are emitted back to front)
    auto rnd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
    static const int steps[8] = {-3, -1, 0, 0, 0, 0, 1, 1};
    for (size_t r = 0; r * L < n; ++r) {
        uint8_t q[512]; int cur = 38;
        for (int i = 0; i < L; ++i) { cur += steps[rnd() & 7]; cur = cur < 2 ? 2 : (cur > 40 ? 40 : cur); q[i] = (uint8_t)cur; }
        const bool rev = (rnd() >> 20) & 1;
        for (int i = 0; i < L && r * L + i < n; ++i) in[r * L + i] = rev ? q[L - 1 - i] : q[i];
    }
    const uint32_t units = fsppmd::SA_SIZE / 12u + 8u;
    g_ser.assign(units, 0); g_win.assign(units, 0);
    uint8_t* arena = (uint8_t*)aligned_alloc(64, (fsppmd::ARENA_BYTES + 4096 + 63) & ~63ull);
    fsppmd::Shared* sh = new fsppmd::Shared;
    uint32_t size = 0;
    simt::run([&](int lane) { uint32_t r0 = 0; const uint32_t r = fsppmd::encode_member(arena, sh, in.data(), (uint32_t)n, out.data(), (uint32_t)out.size(), &r0); if (lane == 0) size = r; });
    printf("symbols %zu -> %u bytes; windows %u covering %u symbols\n", n, size, sh->winStats[1], sh->winStats[2]);
    const uint32_t unitsStart = (1u + fsppmd::SA_SIZE - 12u * (fsppmd::SA_SIZE / 8 / 12 * 7) - 1u) / 12u;
    unsigned long long serTot = 0, winTot = 0, serText = 0;
    for (uint32_t u = 0; u < units; ++u) { serTot += g_ser[u]; winTot += g_win[u]; if (u < unitsStart) serText += g_ser[u]; }
    uint32_t loTouched = 0, hiTouched = units;
    for (uint32_t u = unitsStart; u < units; ++u) if (g_ser[u] | g_win[u]) { loTouched = u; if (false) break; }
    // extent of the two regions
    uint32_t lastList = unitsStart, firstCtx = units - 8;
    { uint32_t gapBest = 0, gapAt = unitsStart, run = 0, runStart = unitsStart;
      for (uint32_t u = unitsStart; u < units - 8; ++u) { if (g_ser[u] | g_win[u]) { if (run > gapBest) { gapBest = run; gapAt = runStart; } run = 0; runStart = u + 1; } else ++run; }
      lastList = gapAt; firstCtx = gapAt + gapBest; }
    printf("heap: text below unit %u (serial accesses there %.1f %%), lists [%u, %u) = %.1f KB, contexts [%u, %u) = %.1f KB\n", unitsStart, 100.0 * serText / serTot,
           unitsStart, lastList, (lastList - unitsStart) * 12 / 1024.0, firstCtx, units - 8, (units - 8 - firstCtx) * 12 / 1024.0);
    printf("accesses: serial walk %llu (wave-uniform), windows %llu (per lane)\n", serTot, winTot);
    for (uint32_t kbLo : {0u, 8u, 16u, 24u, 32u, 48u, 64u, 96u}) for (uint32_t kbHi : {0u, 8u, 16u, 24u, 32u, 48u, 64u}) {
        const uint32_t a = unitsStart + kbLo * 1024u / 12u, b = units - 8 - kbHi * 1024u / 12u;
        unsigned long long s = 0, w = 0;
        for (uint32_t u = unitsStart; u < units; ++u) if (u < a || u >= b) { s += g_ser[u]; w += g_win[u]; }
        printf("  lists %3u KB + contexts %3u KB in LDS: serial %.1f %%  windows %.1f %%\n", kbLo, kbHi, 100.0 * s / serTot, 100.0 * w / winTot);
    }
    return 0;
}
