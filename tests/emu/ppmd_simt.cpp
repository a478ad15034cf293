// TEST-ONLY: the PPMd coder core compiled with its 64-lane code paths (the ones the device runs) on the lock-step
// emulation of tests/emu/simt.h.  build: g++ -O2 -std=c++17 -DFS_SIMT_EMU -shared -fPIC -o build/libsimt_emu.so
//        tests/emu/ppmd_simt.cpp tests/emu/simt.cpp
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
// rescales the short form (packed_rescale_quick) took, each held against the sorting network inside ppmd_window.h
static std::atomic<unsigned long long> g_quickRescales{0};
static inline void simt_count_quick_rescale() { g_quickRescales++; }
extern "C" unsigned long long simt_quick_rescales() { return g_quickRescales.load(); }
#include "simt.h"
#include "../../fastore_amd/csrc/ppmd_core.h"

// the two-wave form: wave 0 walks the model and queues the coding steps, wave 1 is the coder wave; several streams one
// after the other through the same pair (the hand-over between streams is part of what is tested)
extern "C" int simt_ppmd_encode_two_waves(int nStreams, const uint8_t* const* in, const size_t* n, uint8_t* const* out, const size_t* cap, uint32_t* sizes)
{
    uint8_t* arena = (uint8_t*)aligned_alloc(64, (fsppmd::ARENA_BYTES + 4096 + 63) & ~63ull);
    fsppmd::Shared* sh = new fsppmd::Shared;
    sh->qTail = sh->qHead = 0; sh->qStarts = sh->qOpened = 0;
    simt::run_waves(2, [&](int wave, int) {
        if (wave == 1) { fsppmd::coder_wave<false>(sh); return; }
        uint32_t q = 0;
        for (int s = 0; s < nStreams; ++s) {
            uint32_t r0 = 0;
            if (n[s] == 0) { if (simt::lane() == 0) sizes[s] = 0; continue; }
            fsppmd::encode_member(arena, sh, in[s], (uint32_t)n[s], out[s], (uint32_t)cap[s], &r0, true, &sizes[s], q, &q);
        }
        fsppmd::cq_send_exit(sh, q);
    });
    delete sh; free(arena);
    return 0;
}

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
    (void)0;
    if (windowStats) { for (int i = 0; i < 8; ++i) windowStats[i] = sh->winStats[i]; }
    delete sh; free(arena);
    return result;
}

// the windowed form of the small-alphabet range coders (rc_core.h: encode_stream_windowed) on one emulated wave;
// pairs = interleaved (symbol, ctx0) bytes
#include "../../fastore_amd/csrc/rc_core.h"
extern "C" size_t simt_rc_encode(unsigned model, const uint8_t* pairs, size_t n, uint8_t* out, size_t cap)
{
    const uint64_t tb = fsrc::model_table_bytes(model);
    uint8_t* table = (uint8_t*)aligned_alloc(64, (tb + 63) & ~63ull);
    uint32_t size = 0;
    simt::run_waves(1, [&](int, int) {
        const uint32_t s = fsrc::encode_model(model, table, pairs, (uint32_t)n, out, (uint32_t)cap);
        if (simt::lane() == 0) size = s;
    });
    free(table);
    return size == 0xFFFFFFFFu ? (size_t)-1 : size;
}

// the QVZ coder on one emulated wave: form 0 = one symbol per trip (encode_stream), 1 = the windowed form
#include "../../fastore_amd/csrc/qvz_core.h"
extern "C" long simt_qvz_encode(int form, const uint8_t* blob, const uint8_t* syms, size_t n, size_t arenaBytes, uint8_t* out, size_t cap)
{
    uint8_t* arena = (uint8_t*)aligned_alloc(64, (arenaBytes + 4096 + 63) & ~63ull);
    uint32_t size = 0;
    if (form == 2) {      // the two-wave kernel's form: the symbols' counts through the ring, fractions and the interval's pass on the coder wave
        fsppmd::Shared* sh = new fsppmd::Shared;
        sh->qTail = sh->qHead = 0; sh->qStarts = sh->qOpened = 0;
        size = 0xFFFFFFFEu;
        simt::run_waves(2, [&](int wave, int) {
            if (wave == 1) { fsppmd::coder_wave<true, fsqvz::WaveCoder>(sh); return; }
            fsqvz::QvzQueue qq; qq.m.sh = sh; qq.m.qTail = 0; qq.m.qHeadSeen = 0u - fsppmd::CQ_SIZE; qq.sizeOut = &size;
            (void)fsqvz::encode_stream_windowed(arena, blob, syms, (uint32_t)n, out, (uint32_t)cap, &qq);
            fsppmd::cq_send_exit(sh, qq.m.qTail);
        });
        delete sh; free(arena);
        return size == 0xFFFFFFFFu ? -1 : (long)size;
    }
    simt::run_waves(1, [&](int, int) {
        const uint32_t s = form ? fsqvz::encode_stream_windowed(arena, blob, syms, (uint32_t)n, out, (uint32_t)cap) : fsqvz::encode_stream(arena, blob, syms, (uint32_t)n, out, (uint32_t)cap);
        if (simt::lane() == 0) size = s;
    });
    free(arena);
    return size == 0xFFFFFFFFu ? -1 : (long)size;
}


// range-coded streams with their triples coded by the coder wave (two emulated waves), a PPMd member before and behind them:
// the ring carries both kinds of entries, the coder wave switches between its two range coders at the streams' commands
extern "C" int simt_rc_encode_two_waves(int nStreams, const unsigned* models /* 0xFFFFFFFF: a PPMd member */, const uint8_t* const* in, const size_t* n,
                                        uint8_t* const* out, const size_t* cap, uint32_t* sizes)
{
    uint8_t* arena = (uint8_t*)aligned_alloc(64, (size_t)48 << 20);
    fsppmd::Shared* sh = new fsppmd::Shared;
    sh->qTail = sh->qHead = 0; sh->qStarts = sh->qOpened = 0;
    simt::run_waves(2, [&](int wave, int) {
        if (wave == 1) { fsppmd::coder_wave<true>(sh); return; }
        uint32_t q = 0;
        for (int s = 0; s < nStreams; ++s) {
            if (models[s] == 0xFFFFFFFFu) {
                uint32_t r0 = 0;
                if (n[s] == 0) { if (simt::lane() == 0) sizes[s] = 0; continue; }
                fsppmd::encode_member(arena, sh, in[s], (uint32_t)n[s], out[s], (uint32_t)cap[s], &r0, true, &sizes[s], q, &q);
            } else {
                fsrc::RcQueue rq; rq.m.sh = sh; rq.m.qTail = q; rq.m.qHeadSeen = q - fsppmd::CQ_SIZE; rq.sizeOut = &sizes[s];
                (void)fsrc::encode_model_queued(models[s], arena, in[s], (uint32_t)n[s], out[s], (uint32_t)cap[s], &rq);
                q = rq.m.qTail;
            }
        }
        fsppmd::cq_send_exit(sh, q);
    });
    delete sh; free(arena);
    return 0;
}
