// TEST-ONLY: the mate search's kernel body (fastore_amd/csrc/mates_core.h: one wavefront per paired-end bin) on the lock-step
// emulation of tests/emu/simt.h.  Part of build/libsimt_emu.so (tests/test_simt.py); the emulation library's stand-in for
// fs_match_mates calls it when FS_EMU_SIMT_MATES names that library, so that the product's own parity check
// (fsgpu_pe_matcher_check: rows against the host's search) runs over the 64-lane code without a GPU.
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "simt.h"
#if defined(FSM_COUNT)
  #include <atomic>
  static std::atomic<unsigned long long> g_fsm[8];
  #define FSM_STAT(i, x) do { const unsigned long long x_ = (x); if (simt::lane() == 0) g_fsm[i] += x_; } while (0)
  extern "C" void simt_mates_stats(unsigned long long* o) { for (int i = 0; i < 8; ++i) o[i] = g_fsm[i].load(); }
#endif
#include "../../fastore_amd/csrc/mates_core.h"

extern "C" int simt_match_mates(const uint8_t* seq, const fsdev::MatePair* pairs, uint32_t nPairs, const uint32_t* validBits, const fsdev::MateParams* par, fsdev::MateRow* rows)
{
    fsmate::Shared* sh = new fsmate::Shared;
    memset(sh, 0xA5, sizeof *sh);                                            // (nothing may rely on what LDS or the scratch held before)
    std::vector<uint32_t> hist((size_t)fsmate::hist_words(par->window) + 4u, 0xDEADBEEFu);
    uint32_t* h = hist.data(); while (((uintptr_t)h & 15u) != 0u) ++h;
    const fsdev::MateJob job{0u, nPairs};
    simt::run_waves((int)fsmate::kWaves, [&](int, int) { fsmate::search_bin(*sh, job, pairs, seq, validBits, *par, rows, h); });      // (the workgroup: sixteen wavefronts)
    delete sh;
    return 0;
}
