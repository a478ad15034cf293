// Design study (host only): how often each path of the PPMd coder runs and how many memory round trips it makes per
// symbol on quality-like data.  Build: g++ -O2 -std=c++17 -Ifastore_amd/csrc -o build/ppmd_paths tools/ppmd_paths.cpp
#include <stdio.h>
#include <stdlib.h>
#include <random>
#include <vector>
static unsigned long long g_ld[8], g_st, g_path[16], g_region_ops[16], g_region_calls[16];
static int g_region = 0;
#define FS_COUNTERS 1
#include "wave.h"
#include "ppmd_core.h"
int main(int argc, char** argv)
{
    std::mt19937 rng(1); const int n = argc > 1 ? atoi(argv[1]) : 300000; std::vector<uint8_t> in(n);
    const int steps[8] = {-3, -1, 0, 0, 0, 0, 1, 1}; int cur = 38;
    for (int i = 0; i < n; i++) { if (i % 150 == 0) cur = 38; cur += steps[rng() % 8]; if (cur > 40) cur = 40; if (cur < 2) cur = 2; in[i] = (uint8_t)cur; }
    std::vector<uint8_t> out(n + n / 8 + 1024); uint8_t* arena = (uint8_t*)aligned_alloc(64, (fsppmd::ARENA_BYTES + 63) & ~63ull);
    fsppmd::Shared* sh = new fsppmd::Shared; uint32_t rs = 0;
    const uint32_t sz = fsppmd::encode_member(arena, sh, in.data(), n, out.data(), (uint32_t)out.size(), &rs);
    printf("n %d -> %u bytes, %u restarts\n", n, sz, rs);
    const char* names[] = {"symbols", "binary ctx", "multi ctx (sym1)", "sym1 found first", "sym1 found other", "escape levels (sym2)", "sym2 found", "fast path (no update)",
                           "UpdateModel", "CreateSuccessors", "ReduceOrder", "rescale", "update loop iters", "-", "-", "-"};
    for (int i = 0; i < 13; i++) printf("%-24s %10llu  %.3f/sym\n", names[i], g_path[i], (double)g_path[i] / n);
    printf("loads per symbol: ctx %.3f  state %.3f  state-list %.3f  u8 %.3f  u16 %.3f  u32 %.3f  u32h %.3f   | stores %.3f\n", (double)g_ld[0] / n, (double)g_ld[1] / n,
           (double)g_ld[2] / n, (double)g_ld[3] / n, (double)g_ld[4] / n, (double)g_ld[5] / n, (double)g_ld[6] / n, (double)g_st / n);
    const char* regions[] = {"main loop (ctx loads, input)", "encode in first context", "UpdateModel body", "CreateSuccessors", "ReduceOrder", "rescale", "allocator", "allocator rare paths", "encode after escape (sym2)", "StartModel"};
    printf("memory operations by region (host build: a state list costs one operation per state here, one per 64 states on the device)\n");
    for (int i = 0; i < 10; i++) printf("  %-34s calls %9llu  ops %10llu  %.3f/sym  %.1f per call\n", regions[i], g_region_calls[i], g_region_ops[i], (double)g_region_ops[i] / n, g_region_calls[i] ? (double)g_region_ops[i] / g_region_calls[i] : 0.0);
    return 0;
}
