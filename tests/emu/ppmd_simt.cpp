// TEST-ONLY: the PPMd coder core compiled with its 64-lane code paths (the ones the device runs) on the lock-step
// emulation of tests/emu/simt.h.  build: g++ -O2 -std=c++17 -DFS_SIMT_EMU -shared -fPIC -o build/libsimt_emu.so
//        tests/emu/ppmd_simt.cpp tests/emu/simt.cpp
#include <stdlib.h>
#include "../../fastore_amd/csrc/ppmd_core.h"

extern "C" size_t simt_ppmd_encode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, uint32_t* restarts, uint64_t* windowStats)
{
    uint8_t* arena = (uint8_t*)aligned_alloc(64, (fsppmd::ARENA_BYTES + 4096 + 63) & ~63ull);
    fsppmd::Shared* sh = new fsppmd::Shared;
    uint32_t result = 0, rs = 0;
    simt::run([&](int lane) {
        uint32_t r0 = 0;
        const uint32_t r = fsppmd::encode_member(arena, sh, in, (uint32_t)n, out, (uint32_t)cap, &r0);
        if (lane == 0) { result = r; rs = r0; }
    });
    if (restarts) *restarts = rs;
    if (windowStats) { for (int i = 0; i < 8; ++i) windowStats[i] = sh->winStats[i]; }
    delete sh; free(arena);
    return result;
}
