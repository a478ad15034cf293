// TEST-ONLY host emulation of the device entropy-coder cores (one "lane").
// Lets the CPU-only container check bit-exactness of ppmd_core.h / rc_core.h against the oracle
// before a GPU run.  Never loaded by the product.
#include <stdlib.h>
#include "../../fastore_amd/csrc/ppmd_core.h"

extern "C" size_t emu_ppmd_encode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, uint32_t* restarts)
{
    uint8_t* arena = (uint8_t*)aligned_alloc(16, fsppmd::ARENA_BYTES + 16);
    fsppmd::Shared* sh = new fsppmd::Shared;
    uint32_t r = fsppmd::encode_member(arena, sh, in, (uint32_t)n, out, (uint32_t)cap, restarts);
    delete sh; free(arena);
    return r;
}
