// TEST-ONLY: the QVZ coder core compiled for ONE lane (the host form of wave.h) beside the lock-step build of the same header in
// libsimt_emu.so -- the reference the emulated 64-lane forms are held against (tests/test_simt.py).
#include <stdlib.h>
#undef FS_SIMT_EMU
#include "../../fastore_amd/csrc/qvz_core.h"
extern "C" long host_qvz_encode(const uint8_t* blob, const uint8_t* syms, size_t n, size_t arenaBytes, uint8_t* out, size_t cap)
{
    uint8_t* arena = (uint8_t*)aligned_alloc(64, (arenaBytes + 4096 + 63) & ~63ull);
    const uint32_t s = fsqvz::encode_stream(arena, blob, syms, (uint32_t)n, out, (uint32_t)cap);
    free(arena);
    return s == 0xFFFFFFFFu ? -1 : (long)s;
}
