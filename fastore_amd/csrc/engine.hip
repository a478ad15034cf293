// HIP side of the fastore_pack hot path: entropy-coding kernels + the batch engine.
//
// Kernels (gfx950, wave64, one 64-thread workgroup = one wavefront = one stream at a time):
//   fs_encode_streams   work-queue kernel.  Every wave owns a private HBM arena (PPMd heap or
//                       range-coder frequency table) and pulls stream items -- one entropy-coded
//                       stream of one bin -- from a device counter until the queue is empty.
//                       Items are queued longest-first by the host.  No inter-wave hand-off: the
//                       only shared word is the queue head (returning atomicAdd).
//   fs_assemble_blocks  one workgroup per bin: exclusive scan of the stream sizes, block header
//                       (reference layout, big-endian), then a coalesced copy of the streams into
//                       the compact output buffer.
// Reference path replaced: LzCompressorSE::CompressBuffers + StoreHeader
//   (/root/reference/fastore/fastore_pack/FastqCompressor.cpp:1055-1126, 684-699, 1199-1210) and
//   the TEncoder<...>/PpmdEncoder calls behind them.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <sys/mman.h>
#include <unistd.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <stddef.h>
#include <time.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <vector>

#include "device_types.h"
#include "engine.h"
#include "ppmd_core.h"
#include "rc_core.h"
#include "qvz_core.h"
#include "emit_core.h"
#include "emit_wave.h"

using namespace fsdev;

namespace {

constexpr uint64_t kGuard = 64ull << 10;
// Resident coder waves per SIMD.  Measured on MI355X (tools/ppmd_microbench.py): aggregate PPMd throughput saturates
// at ~0.8 G symbols/s from ~3000 waves on (L2 share per wave shrinks), while per-stream latency keeps growing, and a
// launch ends with its longest stream -- so 3 waves/SIMD (3072 waves, no register spills) beats 6.
#ifndef FS_WAVES_PER_SIMD
#define FS_WAVES_PER_SIMD 3
#endif
constexpr uint32_t kWavesPerSimd = FS_WAVES_PER_SIMD;

// Arena slots.  The pool is cut into kXcc partitions of `slotsPerXcc` arenas; a workgroup claims a slot of the XCD it runs
// on and gives it back when its queue is empty.  Slots never migrate between XCDs: the per-XCD L2s are not coherent
// with each other, so an arena reused from another XCD inside one cache epoch could be clobbered by a late write-back
// of the previous owner's dead lines.  A partition is a bitmap (bit set = taken): claim = find a clear bit, atomicOr,
// keep it if the bit was clear before; release = atomicAnd.
// NO workgroup ever waits for another one.  A workgroup that finds its partition full (more workgroups resident on the
// XCD than it has slots: a small pool, e.g. --max-waves or the one-shot pool, under one-wave launches) LEAVES at once, without
// having taken a stream; the streams are taken by the launch's other workgroups, and should none of them have found a slot
// either the host sees the queue's head short of its end and launches again (run_encode: the tail launch).  Rounds 1-4 had
// the workgroup spin for a slot inside the kernel -- a wait for a workgroup of ANOTHER launch, which stands still when that
// launch's hardware queue is not mapped while the spinners hold the compute units (two packs on one device:
// profiles/r04_cli_two_processes.txt).
constexpr uint32_t kXcc = 8;
constexpr uint32_t kBitmapWords = 16;            // 64-bit words per XCD: up to 1024 slots
struct SlotMap { unsigned long long w[kBitmapWords]; };

__device__ __forceinline__ uint32_t xcc_id()
{ uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & (kXcc - 1u); }

// The launch parameters stay in the kernarg segment and are re-read (scalar loads, a few per stream) where they are
// needed: held in SGPRs for the whole kernel they were ~20 registers of pressure on the coder loops, which already
// spill scalars to vector lanes.  `kernargs()` hides the pointer from the optimiser so that the loads are not hoisted.
struct EncodeArgs {
    const StreamItem* items; const uint32_t* order; const uint8_t* in; uint8_t* out; uint32_t* outSizes; uint32_t* restarts;
    uint8_t* arenas; uint64_t arenaStride; uint32_t* queueHead; SlotMap* maps;
    uint32_t nItems, longLen, slotsPerXcc;
    uint32_t budget;      // symbols a workgroup codes in ONE launch before it gives its arena slot back and leaves (0: until the queue is empty)
};
typedef const __attribute__((address_space(4))) EncodeArgs* KernArgs;
__device__ __forceinline__ KernArgs kernargs()
{
    KernArgs k = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return k;
}

// TWO: the two-wave form (128-thread workgroups): wave 0 walks the models and queues every PPMd coding step, wave 1 is
// the coder wave (ppmd_core.h: coder_wave).  Range-coded and QVZ items are coded by wave 0 alone, as in the one-wave form.
// The windowed range coders (rc_core.h) live in kernels of their own (RCWIN: fs_encode_streams_w / fs_encode_streams2_w), taken
// by launches whose range-coded symbols weigh beside their PPMd symbols (a --reduced or --max library's quality scores), and out of line there.
// In the kernels every lossless launch takes they cost the PPMd walk: inlined, nine more vector registers and ten more
// spilled scalars in the two-wave form (a lone 7 M-symbol stream 938 -> 968 ms, the step 5 %); out of line still 2-3 % of
// the step, alternating builds on one box (profiles/r03_reduced_mode.txt).  Those kernels keep the one-symbol loop, which
// is all their short range-coded streams (read ids, flags, letters) need.
__device__ __noinline__ uint32_t rc_encode_out_of_line(uint32_t model, fs_gptr table, fs_cgptr pairs, uint32_t n, fs_gptr out, uint32_t cap)
{ return fsrc::encode_model(model, table, pairs, n, out, cap); }

// (the QVZ coder out of line in every kernel: its double-precision quotients and DPP sums are nothing the PPMd walk's register
// allocation should have to share a function with)
// (FS_QVZ_WINDOWS=0 builds: the one-symbol form, a wave-wide sum and three dependent loads per symbol -- the A/B partner of the windows)
#if !defined(FS_QVZ_WINDOWS)
  #define FS_QVZ_WINDOWS 1
#endif
__device__ __noinline__ uint32_t qvz_encode_out_of_line(fs_gptr arena, fs_cgptr model, fs_cgptr in, uint32_t n, fs_gptr out, uint32_t cap)
{
#if defined(__HIP_DEVICE_COMPILE__) && FS_QVZ_WINDOWS
    return fsqvz::encode_stream_windowed(arena, model, in, n, out, cap);
#else
    return fsqvz::encode_stream(arena, model, in, n, out, cap);      // (and what the compiler's host pass parses)
#endif
}

// (and the one-symbol loop of the range coders: with every coder but PPMd out of line no kernel spills a vector register any more --
// the one-wave kernel had 19-37 spilled and 64-136 bytes of scratch per lane all round; 3 072 equal PPMd streams 6.60 -> 6.86 G symbols/s)
__device__ __noinline__ bool rc_encode_queued_out_of_line(uint32_t model, fs_gptr table, fs_cgptr pairs, uint32_t n, fs_gptr out, uint32_t cap,
                                                          FS_LDS fsppmd::Shared* sh, FS_GLOBAL uint32_t* sizeOut, uint32_t* qTail, uint32_t prio)
{
#if defined(__HIP_DEVICE_COMPILE__)
    fsrc::RcQueue rq; rq.m.sh = sh; rq.m.qTail = *qTail; rq.m.qHeadSeen = *qTail - fsppmd::CQ_SIZE; rq.sizeOut = sizeOut; rq.prio = prio;
    const bool done = fsrc::encode_model_queued(model, table, pairs, n, out, cap, &rq);
    *qTail = rq.m.qTail;
    return done;
#else
    (void)model; (void)table; (void)pairs; (void)n; (void)out; (void)cap; (void)sh; (void)sizeOut; (void)qTail; (void)prio;
    return false;                 // (the host pass of the compiler only parses this)
#endif
}
// (qvz_core.h: a QVZ stream's symbols coded by the coder wave of the two-wave form with the windowed coders)
__device__ __noinline__ void qvz_encode_queued_out_of_line(fs_gptr arena, fs_cgptr model, fs_cgptr in, uint32_t n, fs_gptr out, uint32_t cap,
                                                           FS_LDS fsppmd::Shared* sh, FS_GLOBAL uint32_t* sizeOut, uint32_t* qTail, uint32_t prio)
{
#if defined(__HIP_DEVICE_COMPILE__)
    fsqvz::QvzQueue qq; qq.m.sh = sh; qq.m.qTail = *qTail; qq.m.qHeadSeen = *qTail - fsppmd::CQ_SIZE; qq.sizeOut = sizeOut; qq.prio = prio;
    (void)fsqvz::encode_stream_windowed(arena, model, in, n, out, cap, &qq);
    *qTail = qq.m.qTail;
#else
    (void)arena; (void)model; (void)in; (void)n; (void)out; (void)cap; (void)sh; (void)sizeOut; (void)qTail; (void)prio;      // (the host pass of the compiler only parses this)
#endif
}
__device__ __noinline__ uint32_t rc_serial_out_of_line(uint32_t model, fs_gptr table, fs_cgptr pairs, uint32_t n, fs_gptr out, uint32_t cap)
{ return fsrc::encode_model_serial(model, table, pairs, n, out, cap); }

template <int WAVES, bool RCWIN = false> __device__ __forceinline__ void encode_streams_body()
{
    constexpr bool TWO = WAVES == 2;
    __shared__ fsppmd::Shared sh;
    static_assert(sizeof(fsppmd::Shared) <= 12800, "above 12 800 bytes of LDS a compute unit holds eleven one-wave workgroups, not twelve (profiles/r03_free_list_heads.txt)");
    uint32_t qTail = 0;
    if (TWO) {
        if (threadIdx.x == 0) { sh.qTail = 0u; sh.qHead = 0u; sh.qStarts = 0u; sh.qOpened = 0u; }
        __syncthreads();                               // the workgroup's only barrier: from here on the waves go separate ways
        const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        // (the kernels with the windowed coders: the coder wave also owns the range coder and the QVZ arithmetic coder for the streams whose steps come through the ring)
        if (wv == 1u) { fsppmd::coder_wave<RCWIN, typename std::conditional<RCWIN, fsqvz::WaveCoder, fsppmd::NoQvz>::type>((FS_LDS fsppmd::Shared*)&sh); return; }
    }
    // maps == nullptr: exclusive launch (no other kernel in flight), one arena per workgroup index
    uint32_t slot = blockIdx.x, xcc = 0, word = 0, bit = 0;
    const bool useMaps = kernargs()->maps != nullptr;
    if (useMaps) {
        KernArgs k = kernargs();
        xcc = xcc_id();
        uint32_t s = 0;
        if (threadIdx.x == 0) {
            SlotMap* mp = k->maps + xcc;
            const uint32_t per = k->slotsPerXcc, words = (per + 63u) >> 6;
            uint32_t w0 = blockIdx.x % words;                       // spread the first probes over the words
            for (uint32_t i = 0; i < words && s == 0u; ++i) {
                const uint32_t wi = (w0 + i) % words;
                // bits past the partition's last slot are never offered
                const unsigned long long valid = (wi + 1u) * 64u <= per ? ~0ull : ((1ull << (per - wi * 64u)) - 1ull);
                unsigned long long cur = __hip_atomic_load(&mp->w[wi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while ((~cur & valid) != 0ull) {
                    const uint32_t b = (uint32_t)__builtin_ctzll(~cur & valid);
                    const unsigned long long old = __hip_atomic_fetch_or(&mp->w[wi], 1ull << b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!(old & (1ull << b))) { s = 1u + wi * 64u + b; break; }
                    cur = old | (1ull << b);
                }
            }
        }
        s = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);
        if (s == 0u) {                                  // the partition is full: leave, never wait (the streams stay in the queue for the others, or for the tail launch)
            if (TWO) fsppmd::cq_send_exit((FS_LDS fsppmd::Shared*)&sh, qTail);
            return;
        }
        s -= 1u;
        word = s >> 6; bit = s & 63u;
        slot = xcc * kernargs()->slotsPerXcc + s;
    }
    uint8_t* arena;
    { KernArgs k = kernargs(); arena = k->arenas + (uint64_t)slot * k->arenaStride; }
    uint32_t coded = 0;                                   // symbols this workgroup has coded in this launch (+ a fixed charge per stream)
    for (;;) {
        KernArgs k = kernargs();
        uint32_t q = 0;
        if (threadIdx.x == 0) q = atomicAdd(k->queueHead, 1u);
        q = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
        if (q >= k->nItems) break;                    // every wave reaches this exit: the queue is finite
        const uint32_t it = k->order[q];
        const StreamItem item = k->items[it];
        const uint32_t kind = (uint32_t)__builtin_amdgcn_readfirstlane((int)item.kind);
        const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)item.in_len);
        const uint32_t cap = (uint32_t)__builtin_amdgcn_readfirstlane((int)item.out_cap);
        // The launch ends when its longest stream ends, so the long PPMd streams get issue priority over the thousands
        // of short ones that share their SIMD (priority only reorders issue among resident waves).
        const uint32_t longLen = k->longLen;
        // (a --reduced or --lossy launch: its long streams are range-coded or QVZ ones, and their pass runs on the coder wave, which takes the priority over)
        const bool weighs = kind == KIND_PPMD || kind == KIND_QVZ || kind - KIND_RC_BASE <= fsrc::M_A8O6;
        const uint32_t prio = weighs && n >= longLen ? 3u : (weighs && 2u * n >= longLen ? 2u : 0u);
        if (prio == 3u) __builtin_amdgcn_s_setprio(3);
        else if (prio == 2u) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(0);
        fs_cgptr src = (fs_cgptr)(k->in + item.in_off);
        fs_gptr dst = (fs_gptr)(k->out + item.out_off);
        fs_gptr ar = (fs_gptr)arena;
        uint32_t size = 0, rs = 0;
        bool rcQueued = false;             // (two-wave kernel with the windowed range coders: this stream's size comes from the coder wave)
        const uint64_t tStream = FS_PROF_NOW();
        if (kind == KIND_PPMD) {
            if (n > 0) {
                if (TWO) { KernArgs k3 = kernargs(); (void)fsppmd::encode_member(ar, (FS_LDS fsppmd::Shared*)&sh, src, n, dst, cap, &rs, true, (FS_GLOBAL uint32_t*)(k3->outSizes + it), qTail, &qTail); }
                else size = fsppmd::encode_member(ar, (FS_LDS fsppmd::Shared*)&sh, src, n, dst, cap, &rs);
            }
        } else if (kind == KIND_QVZ) {
            if (TWO && RCWIN && FS_QVZ_WINDOWS) {
                KernArgs k3 = kernargs();
                qvz_encode_queued_out_of_line(ar, (fs_cgptr)(k->in + item.aux_off), src, n, dst, cap, (FS_LDS fsppmd::Shared*)&sh, (FS_GLOBAL uint32_t*)(k3->outSizes + it), &qTail, prio);
                rcQueued = true;
            } else
            size = qvz_encode_out_of_line(ar, (fs_cgptr)(k->in + item.aux_off), src, n, dst, cap);
        } else {
            // (rc_core.h: the small alphabets' triples coded by the coder wave of the two-wave form)
            if (TWO && RCWIN && kind - KIND_RC_BASE <= fsrc::M_A8O6) {
                KernArgs k3 = kernargs();
                rcQueued = rc_encode_queued_out_of_line(kind - KIND_RC_BASE, ar, src, n, dst, cap, (FS_LDS fsppmd::Shared*)&sh, (FS_GLOBAL uint32_t*)(k3->outSizes + it), &qTail, prio);
            } else
            size = RCWIN ? rc_encode_out_of_line(kind - KIND_RC_BASE, ar, src, n, dst, cap) : rc_serial_out_of_line(kind - KIND_RC_BASE, ar, src, n, dst, cap);
        }
        if (threadIdx.x < 16u) {
            KernArgs k2 = kernargs();
            if (threadIdx.x == 0 && !(TWO && kind == KIND_PPMD && n > 0) && !rcQueued) k2->outSizes[it] = size;       // (two-wave form: a PPMd member's size comes from the coder wave)
            // per-stream telemetry: [0] model restarts, [1..6] windowed hit path (attempts, windows, symbols, rounds, redone
            // windows, light rounds), [7] rescales inside windows that let states drop out; FS_WIN_PROFILE builds:
            // [6..7] the serial path's clocks, [8..14] phase clocks / 64 and [15] the stream's whole time / 64
            const uint32_t t = threadIdx.x;
            uint32_t v = 0;
            if (kind == KIND_PPMD && n > 0) {
#if defined(FS_SER_PROFILE)
                if (t >= 1u && t <= 5u) v = sh.serStats[t - 1u];       // (design study: the serial path's clocks in place of the window counters)
                else if (t == 6u || t == 7u) v = sh.winStats[t];
                else if (t >= 8u && t < 15u) v = sh.winStats[t];
#elif defined(FS_WIN_PROFILE)
                if (t >= 1u && t <= 5u) v = sh.winStats[t - 1u];
                else if (t == 6u || t == 7u) v = sh.winStats[t];          // serial-path clocks: escapes, UpdateModel
                else if (t >= 8u && t < 15u) v = sh.winStats[t];
#else
                if (t >= 1u && t <= 6u) v = sh.winStats[t - 1u];
                else if (t == 7u) v = sh.winStats[7];                    // rescales that let states drop out inside windows
#endif
                else if (t == 15u) v = (uint32_t)((FS_PROF_NOW() - tStream) >> 6);
            }
            if (t == 0u) v = rs;
            k2->restarts[16u * it + t] = v;
        }
        FS_WAVE_SYNC();
        // A workgroup's share of one launch is bounded: past it the slot goes back and the workgroup leaves; the streams still in the queue
        // are the next launch's (run_encode launches again while the queue is not empty), whose workgroups take the slots that are free THEN.
        // Without this a launch keeps the workgroups that found a slot in its first microseconds -- a launch made while the pool's other
        // launches held nearly every slot coded its 2 610 streams with a handful of workgroups, 4.7 s instead of 0.25, long after the slots
        // were free again (round 5, tools/cli_stall_hunt.sh).
        if (useMaps) {
            const uint32_t b = kernargs()->budget;
            coded += n > 4096u ? n : 4096u;
            if (b != 0u && coded >= b) break;
        }
    }
    if (TWO) fsppmd::cq_send_exit((FS_LDS fsppmd::Shared*)&sh, qTail);
    if (useMaps && threadIdx.x == 0) {
        KernArgs k = kernargs();
        // the wave's stores have left for the XCD's L2 before the slot shows as free; its next owner runs on this XCD
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_and(&(k->maps + xcc)->w[word], ~(1ull << bit), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(64, kWavesPerSimd) void fs_encode_streams(EncodeArgs /* read through kernargs() */) { encode_streams_body<1>(); }
// (the two-wave form is used where single streams decide the step: two workgroup-waves per SIMD leave each 256 VGPRs -- 3.5 %
// off the time of a lone long stream against the 170 of three per SIMD, profiles/r02_qq_two_waves_per_simd.txt; the one-wave
// form codes the slices of short streams, where the number of resident waves counts)
__global__ __launch_bounds__(128, 2) void fs_encode_streams2(EncodeArgs /* read through kernargs() */) { encode_streams_body<2>(); }
// the same two forms with the windowed range coders (launches with a long range-coded stream)
__global__ __launch_bounds__(64, kWavesPerSimd) void fs_encode_streams_w(EncodeArgs /* read through kernargs() */) { encode_streams_body<1, true>(); }
__global__ __launch_bounds__(128, 2) void fs_encode_streams2_w(EncodeArgs /* read through kernargs() */) { encode_streams_body<2, true>(); }

// fs_gather_quality -- the quality stream of a lossless bin, built on the device (SURVEY 8 a8 + f1): the stored scores
// (.bqua: six bits each, MSB first, fastore_bin/FastqPacker.cpp:157-287) are unpacked, turned back to front where the read
// is stored reverse-complemented (IQualityStoreBase::CompressReadQuality, MET_NONE: FastqCompressor.cpp:229-247 -- the coded
// byte is the score minus the archive's offset, which is exactly the stored six bits) and written in the order in which
// the tree walk emitted the reads, straight into the input place of the bin's PPMd stream.
// One wavefront per string at a time; a lane takes four consecutive OUTPUT symbols: their 24 stored bits lie inside one
// unaligned 32-bit word of the input, and go out as one 32-bit store (the strings follow each other without gaps, so
// consecutive lanes and consecutive strings write consecutive bytes).  HBM-bound: 0.75 B read + 1 B written per score.
__global__ __launch_bounds__(256) void fs_gather_quality(const QuaString* __restrict__ strings, uint32_t nStrings,
                                                         const uint8_t* __restrict__ in, uint8_t* __restrict__ out)
{
    const uint32_t lane = threadIdx.x & 63u, nWaves = gridDim.x * 4u;
    for (uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6); i < nStrings; i += nWaves) {
        const QuaString d = strings[i];
        const uint32_t len = d.len;
        for (uint32_t o = 4u * lane; o < len; o += 256u) {
            const uint32_t cnt = len - o < 4u ? len - o : 4u;
            const uint32_t lo = d.reverse ? len - o - cnt : o;            // lowest stored index among this lane's symbols
            const uint64_t bit = d.src_bit + 6ull * lo;
            uint32_t w; __builtin_memcpy(&w, in + (bit >> 3), 4);
            w = __builtin_bswap32(w) << (uint32_t)(bit & 7u);               // first symbol in the top six bits
            uint32_t s0 = w >> 26, s1 = (w >> 20) & 63u, s2 = (w >> 14) & 63u, s3 = (w >> 8) & 63u;
            if (d.reverse) {                                                // stored order lo .. lo+cnt-1, emitted last first
                const uint32_t a = cnt > 0 ? (cnt == 1 ? s0 : (cnt == 2 ? s1 : (cnt == 3 ? s2 : s3))) : 0u;
                const uint32_t b = cnt == 2 ? s0 : (cnt == 3 ? s1 : s2);
                const uint32_t c = cnt == 3 ? s0 : s1;
                s3 = s0; s0 = a; s1 = b; s2 = c;
            }
            uint8_t* dst = out + d.dst_off + o;
            if (cnt == 4u) { const uint32_t v = s0 | (s1 << 8) | (s2 << 16) | (s3 << 24); __builtin_memcpy(dst, &v, 4); }
            else { dst[0] = (uint8_t)s0; if (cnt > 1u) dst[1] = (uint8_t)s1; if (cnt > 2u) dst[2] = (uint8_t)s2; }
        }
    }
}

// The same for 8-bin / binary archives (IQualityStoreBase::CompressReadQuality, MET_8BIN / MET_BINARY: FastqCompressor.cpp:
// 249-316): the stream is (symbol, context) pairs for the order-k range coder -- symbol = the stored 3-bit bin (1 bit:
// mapped through the archive's threshold), context = emitted index * 8 (2) / length -- and positions under an 'N' base are
// left out, so a lane's pair lands behind the pairs of the earlier positions that stay: its index minus the listed 'N'
// positions in front of it.  One wavefront per string, one position per lane and step, one 16-bit store.
__global__ __launch_bounds__(256) void fs_gather_quality_pairs(const QuaPairString* __restrict__ strings, uint32_t nStrings, const uint8_t* __restrict__ in,
                                                               const uint8_t* __restrict__ nList, uint8_t* __restrict__ out, uint32_t bits, uint32_t sym0, uint32_t sym1)
{
    const uint32_t lane = threadIdx.x & 63u, nWaves = gridDim.x * 4u, ctxMul = bits == 3u ? 8u : 2u;
    for (uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6); s < nStrings; s += nWaves) {
        const QuaPairString d = strings[s];
        const uint32_t len = d.len;
        for (uint32_t i = lane; i < len; i += 64u) {
            const uint32_t ii = d.reverse ? len - 1u - i : i;
            bool isN = false; uint32_t before = 0;
            for (uint32_t t = 0; t < d.n_count; ++t) {
                const uint32_t p = nList[d.n_off + t];
                isN = isN || p == ii;
                before += d.reverse ? (p > ii ? 1u : 0u) : (p < ii ? 1u : 0u);
            }
            if (isN) continue;
            const uint64_t bit = d.src_bit + (uint64_t)bits * ii;
            const uint32_t w = ((uint32_t)in[bit >> 3] << 8) | in[(bit >> 3) + 1];
            uint32_t sym = (w >> (16u - bits - (uint32_t)(bit & 7u))) & ((1u << bits) - 1u);
            if (bits == 1u) sym = sym ? sym1 : sym0;
            const uint32_t ctx = i * ctxMul / len;
            const uint16_t v = (uint16_t)(sym | (ctx << 8));
            __builtin_memcpy(out + 2ull * (d.dst_off + i - before), &v, 2);
        }
    }
}

// fs_gather_quality_qvz -- the quality stream of a --lossy (QVZ) bin, built on the device (SURVEY 8 a8 + f1): per score the
// conditional quantizer is chosen (column = emitted index, the previous QUANTIZED value, and a 7-bit draw of the archive's
// WELL-1024a generator against the pair's ratio: choose_quantizer, fastore_pack/quantizer.cpp:522-531), the score is quantized and
// the place of the quantized value in that quantizer's output alphabet goes to the arithmetic coder with the quantizer's number
// as its context (IQualityStoreBase::CompressReadQuality, MET_QVZ: FastqCompressor.cpp:318-364).  A read is a chain -- every
// choice needs the quantized value before it -- but reads are independent, and so are the draws: the generator is re-seeded at
// every bin and draws once per score in emission order, so draw number d of ANY bin is bits [7 (d mod 4), +7) of its output
// number d / 4 (well_1024a_bits keeps 28 of an output's 32 bits, well.cpp:42-55), a table the host makes once per archive.
// One read per lane.  A read the codebook cannot take (longer than its
// columns, a previous value outside the column's alphabet, a score its quantizer has no entry for) gets the word 0xFFFFFFFF,
// which the coder kernel answers with the stream's error -- as the host's symbolisation throws.
__global__ __launch_bounds__(256) void fs_gather_quality_qvz(const QuaQvzString* __restrict__ strings, uint32_t nStrings, const uint8_t* __restrict__ in, uint8_t* __restrict__ out)
{
    for (uint32_t s = blockIdx.x * 256u + threadIdx.x; s < nStrings; s += gridDim.x * 256u) {
        const QuaQvzString d = strings[s];
        const uint8_t* model = in + 16ull * d.model16;
        const QvzSymHeader* h = (const QvzSymHeader*)model;
        const uint32_t columns = h->columns, nCtx = h->n_ctx, wellWords = h->well_words;
        const uint32_t* colCtxBase = (const uint32_t*)(model + h->col_ctx_base_off);
        const uint16_t* colIndex = (const uint16_t*)(model + h->col_index_off);
        const uint8_t* qratio = model + h->qratio_off; const uint8_t* quant = model + h->quant_off; const uint8_t* stateOf = model + h->state_of_off;
        const uint32_t* well = (const uint32_t*)(model + h->well_off);
        uint32_t* dst = (uint32_t*)(out + d.dst_off);
        const uint32_t len = d.len;
        uint32_t prev = 0; bool bad = len > columns;
        for (uint32_t i = 0; i < len; ++i) {
            uint32_t w = 0xFFFFFFFFu;
            if (!bad) {
                const uint32_t ii = d.reverse ? len - 1u - i : i;
                const uint64_t bit = d.src_bit + 6ull * ii;
                const uint32_t two = ((uint32_t)in[bit >> 3] << 8) | in[(bit >> 3) + 1];
                const uint32_t qv = (two >> (10u - (uint32_t)(bit & 7u))) & 63u;
                const uint32_t idx = prev < 82u ? colIndex[i * 82u + prev] : 0xFFFFu;
                const uint32_t dn = d.draw0 + i;
                if (idx == 0xFFFFu || (dn >> 2) >= wellWords) bad = true;
                else {
                    const uint32_t pair = colCtxBase[i] / 2u + idx;
                    const uint32_t draw = (well[dn >> 2] >> (7u * (dn & 3u))) & 127u;
                    const uint32_t ctx = 2u * pair + (draw >= qratio[pair] ? 1u : 0u);
                    if (ctx >= nCtx) bad = true;
                    else {
                        const uint32_t st = stateOf[ctx * 72u + qv];
                        if (st == 0xFFu) bad = true;
                        else { w = ctx | (st << 24); prev = quant[ctx * 72u + qv]; }
                    }
                }
            }
            dst[i] = w;
        }
    }
}

// fs_tokenise_ids -- the read-id streams of a bin, built on the device (SURVEY 8 a10 + f1): the stored headers (.bhead: 7 bits per
// character behind an implied '@', fastore_bin/FastqPacker.cpp:157-287) are split at each field's own separator, constant
// fields skipped, token fields looked up in the field's value list (their index, context = field), numeric fields written
// as the big-endian bytes of (value - minimum) with context = field * 4 + byte (IHeaderStoreBase::CompressReadId,
// fastore_pack/FastqCompressor.cpp:504-583).  One workgroup per bin, one read per thread and step; a read's pairs land behind
// those of the reads before it: counted first, placed by a prefix sum over the workgroup, then written.  The two streams'
// lengths go to their stream items, which the coder kernel reads behind this one.
// (Round 3: a bin is cut into chunks of 256 reads, one workgroup each -- a 47 000-read bin used to be 186 rounds of ONE workgroup
// on ONE compute unit, 50-330 ms in front of the slice's coder launch on the same stream.  fs_id_count parses every read once
// and leaves its pair counts and the chunk's totals; fs_id_write places a chunk behind the totals of the bin's earlier chunks
// and parses again to write.  Characters come eight at a time: 56 stored bits from one unaligned 64-bit load.)
struct IdChunk { uint32_t job, c0, first_chunk, pad; };          // reads [c0, c0 + 256) of the job; the job's first chunk
// A read id's stored characters -- seven bits each from bit src_bit on -- lie inside the uploaded input.  IdChars::get refills with
// ONE unaligned 64-bit load at the byte of the character it needs, so its last load of a string can reach seven bytes past the
// string's last byte; those bytes are never used, and they are always readable: every buffer that holds the input (dIn) is made
// kInSlack bytes longer than anything addressed in it (the `+ kInSlack` of every ensure(dev->dIn ...) below).
constexpr uint32_t kInSlack = 64;
static_assert(kInSlack >= 8, "IdChars::get reads eight bytes at a time");
static inline bool id_string_inside(uint64_t src_bit, uint32_t len, uint64_t inputBytes) { return (src_bit + 7ull * len + 7ull) / 8ull <= inputBytes; }
struct IdChars {
    const uint8_t* in; uint64_t src_bit; uint64_t buf; uint32_t first;      // stored characters [first, first + 8) sit in the top 56 bits of buf
    __device__ __forceinline__ uint32_t get(uint32_t j)
    {
        if (j == 0u) return (uint32_t)'@';
        const uint32_t i = j - 1u;
        if (i - first >= 8u) {
            const uint64_t bit = src_bit + 7ull * i;
            uint64_t w; __builtin_memcpy(&w, in + (bit >> 3), 8);
            buf = __builtin_bswap64(w) << (uint32_t)(bit & 7u);
            first = i;
        }
        return (uint32_t)(buf >> (57u - 7u * (i - first))) & 127u;
    }
};
template <bool WRITE> __device__ __forceinline__ void id_parse(const IdString& s, const uint8_t* __restrict__ in, const uint8_t* __restrict__ tab, uint32_t& nt, uint32_t& nv,
                                                               uint8_t* tokDst, uint8_t* valDst)
{
    const uint32_t nf = *(const uint32_t*)tab;
    const IdField* F = (const IdField*)(tab + 8);
    IdChars scan{in, s.src_bit, 0ull, 0xFFFFFF00u}, look{in, s.src_bit, 0ull, 0xFFFFFF00u};      // the separator scan, and the look back into a field
    uint32_t fieldStart = 0, fi = 0;
    for (uint32_t i = 0; i <= s.len; ++i) {
        if (fi >= nf) break;
        const IdField f = F[fi];
        if (i != s.len && scan.get(i) != (uint32_t)f.separator) continue;
        if (!f.is_const) {
            const uint32_t fieldLen = i - fieldStart;
            if (!f.is_numeric) {
                uint32_t id = f.n_values;                                  // std::find's end() when the value is not listed
                const uint32_t* vl = (const uint32_t*)(tab + f.values_off);
                for (uint32_t v = 0; v < f.n_values && id == f.n_values; ++v) {
                    if (vl[2 * v + 1] != fieldLen) continue;
                    const uint8_t* vb = tab + vl[2 * v];
                    bool same = true;
                    for (uint32_t k = 0; k < fieldLen && same; ++k) same = vb[k] == look.get(fieldStart + k);
                    if (same) id = v;
                }
                if (WRITE) { tokDst[2 * nt] = (uint8_t)id; tokDst[2 * nt + 1] = (uint8_t)fi; }
                ++nt;
            } else {
                uint64_t v = 0;
                for (uint32_t k = 0; k < fieldLen; ++k) { const uint32_t c = look.get(fieldStart + k); if (c < '0' || c > '9') break; v = v * 10u + (c - '0'); }
                const int64_t diff = (int64_t)(v - f.min_value);
                uint32_t ctx = fi << 2;
                for (int32_t p = (int32_t)f.plog; p >= 0; --p) {
                    if (WRITE) { valDst[2 * nv] = (uint8_t)((diff >> (8 * p)) & 0xFF); valDst[2 * nv + 1] = (uint8_t)ctx; }
                    ++ctx; ++nv;
                }
            }
        }
        fieldStart = i + 1u; ++fi;
    }
}

// sums of (a, b) over the 256 threads of the workgroup (every thread gets both)
__device__ __forceinline__ void id_block_sum(uint32_t& a, uint32_t& b, uint32_t* sA, uint32_t* sB)
{
    const uint32_t tid = threadIdx.x;
    sA[tid] = a; sB[tid] = b;
    __syncthreads();
    for (uint32_t d = 128u; d > 0u; d >>= 1) {
        if (tid < d) { sA[tid] += sA[tid + d]; sB[tid] += sB[tid + d]; }
        __syncthreads();
    }
    a = sA[0]; b = sB[0];
    __syncthreads();
}

__global__ __launch_bounds__(256) void fs_id_count(const IdChunk* __restrict__ chunks, const IdJob* __restrict__ jobs, const IdString* __restrict__ strings, const uint8_t* __restrict__ in,
                                                   uint32_t* __restrict__ counts, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t sT[256], sV[256];
    const IdChunk ck = chunks[blockIdx.x];
    const IdJob job = jobs[ck.job];
    const uint32_t tid = threadIdx.x;
    uint32_t nt = 0, nv = 0;
    if (ck.c0 + tid < job.count) {
        const IdString s = strings[job.first + ck.c0 + tid];
        id_parse<false>(s, in, in + job.table_off, nt, nv, nullptr, nullptr);
        counts[job.first + ck.c0 + tid] = nt | (nv << 16);          // (a read id has at most 255 characters: both counts stay far below 65 536)
    }
    id_block_sum(nt, nv, sT, sV);
    if (tid == 0u) { totals[2u * blockIdx.x] = nt; totals[2u * blockIdx.x + 1u] = nv; }
}

__global__ __launch_bounds__(256) void fs_id_write(const IdChunk* __restrict__ chunks, const IdJob* __restrict__ jobs, const IdString* __restrict__ strings, const uint8_t* __restrict__ in,
                                                   const uint32_t* __restrict__ counts, const uint32_t* __restrict__ totals, uint8_t* __restrict__ out, StreamItem* items)
{
    __shared__ uint32_t sT[256], sV[256];
    const IdChunk ck = chunks[blockIdx.x];
    const IdJob job = jobs[ck.job];
    const uint32_t tid = threadIdx.x;
    // pairs of the bin's earlier chunks
    uint32_t tokBase = 0, valBase = 0;
    for (uint32_t c = ck.first_chunk + tid; c < blockIdx.x; c += 256u) { tokBase += totals[2u * c]; valBase += totals[2u * c + 1u]; }
    id_block_sum(tokBase, valBase, sT, sV);
    const bool active = ck.c0 + tid < job.count;
    uint32_t nt = 0, nv = 0;
    if (active) { const uint32_t c = counts[job.first + ck.c0 + tid]; nt = c & 0xFFFFu; nv = c >> 16; }
    // inclusive prefix sums over the workgroup (Hillis-Steele in LDS)
    sT[tid] = nt; sV[tid] = nv;
    __syncthreads();
    for (uint32_t d = 1; d < 256u; d <<= 1) {
        const uint32_t a = tid >= d ? sT[tid - d] : 0u, b = tid >= d ? sV[tid - d] : 0u;
        __syncthreads();
        sT[tid] += a; sV[tid] += b;
        __syncthreads();
    }
    const uint32_t offT = sT[tid] - nt, offV = sV[tid] - nv, totT = sT[255], totV = sV[255];
    if (active) {
        const IdString s = strings[job.first + ck.c0 + tid];
        uint32_t wt = 0, wv = 0;
        id_parse<true>(s, in, in + job.table_off, wt, wv, out + job.tok_out + 2ull * (tokBase + offT), out + job.val_out + 2ull * (valBase + offV));
    }
    // the bin's last chunk knows the streams' lengths: they go to their stream items, which the coder kernel reads behind this one
    if (tid == 0u && ck.c0 + 256u >= job.count) { items[job.tok_item].in_len = tokBase + totT; items[job.val_item].in_len = valBase + totV; }
}


// ---- device-side stream emission (SURVEY 8 a6 + a11): fsdev::EmitOp, emit_core.h ----
// The streams of a bin that hold bases are written here from the ops the host's walk left.  A WAVEFRONT per op (emit_wave.h: 64 lanes
// take 64 consecutive positions of the record, a letter's place is a population count of a ballot, the match bits ARE a ballot and land
// in a packed word array), the op's walk twice (fs_emit_count, then fs_emit_write behind the counts of the ops in front of it in ITS
// channels: fs_emit_scan, a workgroup per bin); the packed match bits of the run-length coded channels go through fs_rle_binary, the
// bin's LZ ids through fs_rle0 -- BinaryRleEncoder and Rle0Encoder (rle/RleEncoder.h:21-79, 140-212) as block scans: what a position
// emits depends on the symbols in front of it only through the length of the run it stands in (mod 253: a byte 255 is due every 253
// ones; mod 2: zeros go in pairs), and "the run so far" is a scan of maps that either restart the count (a zero / a value) or add to it.
// The block scans: inside a wavefront by lane shuffles (six steps, no LDS, no barrier), across the workgroup's waves through one
// LDS word a wave and two barriers (round 4: Hillis-Steele over 256 LDS words, sixteen barriers a scan).
template <int T> __device__ __forceinline__ uint32_t block_exclusive_sum(uint32_t v, uint32_t* sm, uint32_t& total)
{
    static_assert(T % 64 == 0 && T <= 1024, "whole wavefronts");
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t x = v;
    #pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, (int)d, 64); if (lane >= d) x += y; }
    if (lane == 63u) sm[wave] = x;
    __syncthreads();
    uint32_t before = 0, all = 0;
    #pragma unroll
    for (uint32_t w = 0; w < (uint32_t)T / 64u; ++w) { const uint32_t t = sm[w]; if (w < wave) before += t; all += t; }
    total = all;
    __syncthreads();
    return before + x - v;
}
// Every thread holds a map of a counter: `restart` -> the counter becomes v, else -> (counter + v) mod M.  Returns the counter as
// it stands IN FRONT of this thread when `carry` stands in front of the block; carryOut: behind the block's last thread.
// (maps compose: (k2, v2) after (k1, v1) = k2 ? (1, v2) : (k1, (v1 + v2) mod M))
template <int T> __device__ __forceinline__ uint32_t block_counter_scan(bool restart, uint32_t v, uint32_t M, uint32_t carry, uint32_t* sk, uint32_t* sv, uint32_t& carryOut)
{
    static_assert(T % 64 == 0 && T <= 1024, "whole wavefronts");
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t k = restart ? 1u : 0u, x = v % M;
    #pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t pk = (uint32_t)__shfl_up((int)k, (int)d, 64), px = (uint32_t)__shfl_up((int)x, (int)d, 64);
        if (lane >= d && !k) { k = pk; x = (px + x) % M; }
    }
    // the map of the lanes in front of me inside my wave (lane 0: none)
    const uint32_t ek = (uint32_t)__shfl_up((int)k, 1, 64), ex = (uint32_t)__shfl_up((int)x, 1, 64);
    if (lane == 63u) { sk[wave] = k; sv[wave] = x; }
    __syncthreads();
    // the counter in front of my wave: the carry through the maps of the waves in front; and behind the block: through all of them
    uint32_t cur = carry, mine = carry;
    #pragma unroll
    for (uint32_t w = 0; w < (uint32_t)T / 64u; ++w) { if (w == wave) mine = cur; cur = sk[w] ? sv[w] : (cur + sv[w]) % M; }
    carryOut = cur;
    __syncthreads();
    return lane == 0u ? mine : (ek ? ex : (mine + ex) % M);
}

// one wavefront per op: what it writes into its (at most two) channels
__device__ __forceinline__ void emit_count_body(const EmitJob* jobs, const EmitOp* ops, uint32_t nOps, const uint8_t* in, uint32_t* counts)
{
    const uint32_t g = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (g >= nOps) return;                                              // (the whole wavefront)
    const EmitOp op = ops[g];
    const EmitJob& job = jobs[op.pad2[0]];
    const fsemit::WaveCount c = fsemit::emit_op_wave<false>(op, job, in + job.seq_off, in + job.contig_off, nullptr, nullptr, nullptr, 0u);
    if ((threadIdx.x & 63u) == 0u) { counts[2u * g] = c.nL; counts[2u * g + 1u] = c.nB; }
}
__global__ __launch_bounds__(256) void fs_emit_count(const EmitJob* __restrict__ jobs, const EmitOp* __restrict__ ops, uint32_t nOps, const uint8_t* __restrict__ in, uint32_t* __restrict__ counts)
{ emit_count_body(jobs, ops, nOps, in, counts); }

// one workgroup per bin: every op's place in its channels; the channels' totals; the lengths of the streams that are complete with
// fs_emit_write (all but the run-length coded ones) into their stream items; the packed words of the bit channels cleared (fs_emit_write
// ORs the match bits into them)
__global__ __launch_bounds__(256) void fs_emit_scan(const EmitJob* __restrict__ jobs, const EmitOp* __restrict__ ops, const uint32_t* __restrict__ counts, uint32_t* __restrict__ offs,
                                                    uint32_t* __restrict__ totals, StreamItem* items, uint8_t* __restrict__ out)
{
    __shared__ uint32_t sm[4];
    __shared__ uint32_t run[ECH_COUNT];
    const EmitJob& job = jobs[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    if (tid < ECH_COUNT) run[tid] = 0u;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < job.n_ops; c0 += 256u) {
        const bool active = c0 + tid < job.n_ops;
        const uint32_t g = job.first_op + c0 + tid;
        uint32_t chL = ECH_COUNT, chB = ECH_COUNT, nL = 0, nB = 0;
        if (active) { const EmitOp op = ops[g]; chL = fsemit::channel_l(op); chB = fsemit::channel_b(op); nL = counts[2u * g]; nB = counts[2u * g + 1u]; }
        uint32_t offL = 0, offB = 0;
        for (uint32_t c = 0; c < ECH_COUNT; ++c) {
            // (a channel is either an L or a B channel, never both)
            const uint32_t mine = chL == c ? nL : (chB == c ? nB : 0u);
            if (__syncthreads_or(mine != 0u) == 0) continue;
            uint32_t total;
            const uint32_t ex = block_exclusive_sum<256>(mine, sm, total);
            const uint32_t base = run[c];
            if (chL == c) offL = base + ex;
            if (chB == c) offB = base + ex;
            __syncthreads();
            if (tid == 0u) run[c] = base + total;
            __syncthreads();
        }
        if (active) { offs[2u * g] = offL; offs[2u * g + 1u] = offB; }
    }
    __syncthreads();
    if (tid < ECH_COUNT) {
        totals[(ECH_COUNT + 1u) * blockIdx.x + tid] = run[tid];
        if (!fsemit::is_bit_channel(tid) && job.item[tid] != 0xFFFFFFFFu) items[job.item[tid]].in_len = run[tid];
    }
    for (uint32_t c = 0; c < ECH_COUNT; ++c) {
        if (!fsemit::is_bit_channel(c) || job.item[c] == 0xFFFFFFFFu) continue;
        uint32_t* words = (uint32_t*)(out + job.raw_off[c]);
        const uint32_t n = (run[c] + 31u) / 32u + 2u;                      // (+ 2: append_bits may touch two words behind the last bit's; the room is the channel's cap in BYTES: >= n * 4)
        for (uint32_t i = tid; i < n; i += 256u) words[i] = 0u;
    }
}

__device__ __forceinline__ void emit_write_body(const EmitJob* jobs, const EmitOp* ops, uint32_t nOps, const uint8_t* in, const uint32_t* offs, uint8_t* out)
{
    const uint32_t g = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (g >= nOps) return;
    const EmitOp op = ops[g];
    const EmitJob& job = jobs[op.pad2[0]];
    const uint32_t chL = fsemit::channel_l(op), chB = fsemit::channel_b(op);
    const uint32_t offL = offs[2u * g], offB = offs[2u * g + 1u];
    uint8_t* outL = chL < ECH_COUNT ? out + job.out_off[chL] + (uint64_t)fsemit::unit_l(chL) * offL : nullptr;
    const bool bitCh = chB < ECH_COUNT && fsemit::is_bit_channel(chB);
    uint8_t* outSym = (chB < ECH_COUNT && !bitCh) ? out + job.out_off[chB] + 2ull * offB : nullptr;
    uint32_t* outBits = bitCh ? (uint32_t*)(out + job.raw_off[chB]) : nullptr;
    (void)fsemit::emit_op_wave<true>(op, job, in + job.seq_off, in + job.contig_off, outL, outSym, outBits, offB);
}
__global__ __launch_bounds__(256) void fs_emit_write(const EmitJob* __restrict__ jobs, const EmitOp* __restrict__ ops, uint32_t nOps, const uint8_t* __restrict__ in, const uint32_t* __restrict__ offs,
                                                     uint8_t* __restrict__ out)
{ emit_write_body(jobs, ops, nOps, in, offs, out); }

// BinaryRleEncoder over one bit channel of one bin (blockIdx.x = 3 * bin + which).  The bits come packed (LSB first: fs_emit_write);
// a thread takes 16 of them with one shift; a chunk is 4 096 bits.
__global__ __launch_bounds__(256) void fs_rle_binary(const EmitJob* __restrict__ jobs, const uint32_t* __restrict__ totals, uint8_t* out, StreamItem* items)
{
    __shared__ uint32_t sk[4], sv[4];
    const uint32_t chOf[3] = {ECH_MATCH_BITS, ECH_CMATCH_BITS, ECH_MATCH_BITS_PE};
    const uint32_t j = blockIdx.x / 3u, ch = chOf[blockIdx.x % 3u];
    const EmitJob& job = jobs[j];
    if (job.item[ch] == 0xFFFFFFFFu) return;                              // (uniform: the whole workgroup)
    const uint32_t n = totals[(ECH_COUNT + 1u) * j + ch], tid = threadIdx.x;
    const uint32_t* words = (const uint32_t*)(out + job.raw_off[ch]);
    uint8_t* dst = out + job.out_off[ch];
    uint32_t carry = 0, written = 0;
    for (uint32_t base = 0; base < n; base += 4096u) {
        const uint32_t i0 = base + 16u * tid, cnt = i0 >= n ? 0u : (n - i0 < 16u ? n - i0 : 16u);
        const uint32_t m = cnt ? (words[i0 >> 5] >> (i0 & 31u)) & ((1u << cnt) - 1u) : 0u;      // bit k: position i0 + k holds a one
        const uint32_t zeros = cnt ? ~m & ((1u << cnt) - 1u) : 0u;
        const uint32_t lead = zeros ? (uint32_t)__builtin_ctz(zeros) : cnt;              // ones in front of my first zero
        const uint32_t tail = zeros ? cnt - 1u - (31u - (uint32_t)__builtin_clz(zeros)) : 0u;      // ones behind my last zero
        uint32_t carryOut;
        const uint32_t before = block_counter_scan<256>(zeros != 0u, zeros ? tail : cnt, 253u, carry, sk, sv, carryOut);
        // my bytes: one per zero, and one 255 if the run I continue reaches 253 inside my leading ones
        const uint32_t mine = (uint32_t)__popc(zeros) + ((lead > 0u && before + lead >= 253u) ? 1u : 0u);
        uint32_t total;
        const uint32_t at = written + block_exclusive_sum<256>(mine, sk, total);
        uint32_t cur = before, w = at;
        for (uint32_t k = 0; k < cnt; ++k) {
            if ((m >> k) & 1u) { if (++cur == 253u) { dst[w++] = 255u; cur = 0u; } }
            else { dst[w++] = cur ? (uint8_t)(cur + 2u) : (uint8_t)0; cur = 0u; }
        }
        carry = carryOut; written += total;
    }
    if (tid == 0u) {
        if (carry) dst[written++] = (uint8_t)(carry + 2u);
        items[job.item[ch]].in_len = written;
    }
}

// Rle0Encoder over the LZ ids of one bin.  A thread takes one id; the counter is the parity of the zeros in front of it.
__global__ __launch_bounds__(256) void fs_rle0(const EmitJob* __restrict__ jobs, const uint32_t* __restrict__ ids, uint8_t* out, StreamItem* items)
{
    __shared__ uint32_t sk[4], sv[4];
    const EmitJob& job = jobs[blockIdx.x];
    if (job.item[ECH_COUNT] == 0xFFFFFFFFu) return;
    const uint32_t n = job.n_ids, tid = threadIdx.x;
    uint8_t* dst = out + job.out_off[ECH_COUNT];
    uint32_t carry = 0, written = 0;
    for (uint32_t base = 0; base < n; base += 256u) {
        const bool active = base + tid < n;
        const uint32_t v = active ? ids[job.first_id + base + tid] : 0u;
        uint32_t carryOut;
        // a value restarts the count of zeros at 0; a zero adds one (mod 2); a thread past the end leaves it as it is
        const uint32_t odd = block_counter_scan<256>(active && v != 0u, (active && v == 0u) ? 1u : 0u, 2u, carry, sk, sv, carryOut);
        const uint32_t mine = !active ? 0u : (v == 0u ? (odd ? 1u : 0u) : (odd ? 1u : 0u) + fsemit::rle0_value_bytes(v));
        uint32_t total;
        uint32_t w = written + block_exclusive_sum<256>(mine, sk, total);
        if (active) {
            if (v == 0u) { if (odd) dst[w] = 0u; }
            else { if (odd) dst[w++] = 1u; (void)fsemit::rle0_put_value(v, dst + w); }
        }
        carry = carryOut; written += total;
    }
    if (tid == 0u) {
        if (carry) dst[written++] = 1u;
        items[job.item[ECH_COUNT]].in_len = written;
    }
}

__device__ __forceinline__ void put_be(uint8_t* p, uint64_t v, int nbytes)
{ for (int i = 0; i < nbytes; ++i) p[i] = (uint8_t)(v >> (8 * (nbytes - 1 - i))); }

__global__ __launch_bounds__(256) void fs_assemble_blocks(const BlockPlan* __restrict__ plans, const StreamItem* __restrict__ items,
                                                          const uint32_t* __restrict__ sizes, const uint8_t* __restrict__ scratch,
                                                          uint8_t* blocks)
{
    const BlockPlan& pl = plans[blockIdx.x];
    const uint32_t N = pl.n_streams;
    __shared__ uint64_t dstOff[MAX_STREAMS];
    __shared__ uint64_t payloadEnd;
    uint8_t* blk = blocks + pl.block_off;
    const uint64_t headerSize = 42ull + 16ull * N;
    if (threadIdx.x == 0) {
        uint64_t pos = headerSize;
        for (uint32_t k = 0; k < N; ++k) { const uint32_t s = pl.copy_order[k]; dstOff[s] = pos; pos += sizes[pl.first_item + s]; }
        payloadEnd = pos;
        // StoreRawHeader + work sizes + comp sizes (FastqCompressor.cpp:56-69, 684-699)
        uint8_t* h = blk;
        put_be(h, pl.signature, 4); h += 4;
        put_be(h, pl.records, 8); h += 8;
        *h++ = pl.min_len; *h++ = pl.max_len;
        put_be(h, pl.raw_dna_size, 8); h += 8;
        put_be(h, pos, 8); h += 8;                    // footerOffset
        put_be(h, 1, 4); h += 4;                      // footerSize
        if (pl.has_headers) { put_be(h, pl.raw_id_size, 8); h += 8; }
        // (range-coded streams: work size == coded size; streams the device wrote itself, ~1: their length as its kernels left it)
        for (uint32_t s = 0; s < N; ++s) { const uint64_t ws = pl.work_size[s]; put_be(h, ws == ~0ull ? (uint64_t)sizes[pl.first_item + s] : (ws == ~1ull ? (uint64_t)items[pl.first_item + s].in_len : ws), 8); h += 8; }
        for (uint32_t s = 0; s < N; ++s) { put_be(h, sizes[pl.first_item + s], 8); h += 8; }
        while (h < blk + headerSize) *h++ = 0;        // the 8-byte hole of header-less archives: defined as zero here
        blk[pos] = 0;                                 // footer: sampleValue
    }
    __syncthreads();
    for (uint32_t s = 0; s < N; ++s) {
        const uint32_t n = sizes[pl.first_item + s];
        const uint8_t* src = scratch + items[pl.first_item + s].out_off;
        uint8_t* dst = blk + dstOff[s];
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    }
}

// Waiting for a lane's stream: an event that is POLLED between sleeps, neither hipStreamSynchronize nor a blocking wait.  The runtime's
// default wait spins on a core for the whole device call (with fourteen lanes that is fourteen cores taken from the host front end), and a
// hipEventSynchronize on a blocking-sync event has been seen to sleep THROUGH the end of the work it waits for: one fastore_pack e process in ten
// to twenty, behind processes that had just given 18 GB back, sat 8-25 s in front of an idle stream -- queue drained, no arena slot taken -- until
// something else woke it (round 4's "one stall that is not explained", round 5's tools/cli_stall_hunt.sh with FS_WATCHDOG).  Sleeps of 20 us
// doubling to 1 ms: a long kernel costs a lane thread a thousand wake-ups a second, the end of the work is seen within a millisecond.
static hipError_t wait_event_polled(hipEvent_t ev)
{
    long ns = 20000;
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        struct timespec ts = {0, ns}; nanosleep(&ts, nullptr);
        if (ns < 1000000L) ns *= 2;
    }
}
static hipError_t wait_stream(fsengine::Device* dev, hipStream_t st)
{
    hipError_t e = hipEventRecord((hipEvent_t)dev->evWait, st);
    if (e != hipSuccess) return e;
    return wait_event_polled((hipEvent_t)dev->evWait);
}

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(dev->err, sizeof dev->err, "%s failed: %s", #x, hipGetErrorString(e_)); return -1; } } while (0)

template <class T> int ensure(fsengine::Device* dev, T*& p, size_t& capBytes, size_t needBytes)
{
    if (needBytes <= capBytes && p) return 0;
    if (p) { if (dev->nOldDev < 32u) dev->oldDev[dev->nOldDev++] = (void*)p; else (void)hipFree(p); }      // (kept until nothing is in flight: engine.h)
    p = nullptr; capBytes = 0;
    size_t want = needBytes + needBytes / 4 + 4096;
    HIP_TRY(hipMalloc((void**)&p, want));
    capBytes = want;
    return 0;
}

}  // namespace

namespace fsengine {

// The lanes of a context are HIP streams whose kernels must be able to run side by side.  The runtime multiplexes
// streams onto 4 hardware queues by default, and streams sharing a queue run their kernels one after the other
// (measured: 5 lanes on 4 queues lose 18 % of the step).  Takes effect only if HIP is not initialised yet; a process
// that initialises HIP first (PyTorch) sets the variable itself (bench.py does).
static void want_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

int device_count()
{
    want_hw_queues();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static double wallMs() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6; }

// Shared by the lanes of one GPU.  `gate`: launches that take their arenas from the slot maps hold it shared; a launch
// whose coder table does not fit a slot (the sparse <256,1> read-id model of archives with > 16 header fields) runs
// alone, with one arena per workgroup index carved at its own stride.
struct Pool {
    uint8_t* arenas = nullptr; uint64_t bytes = 0, slotStride = 0;
    uint32_t slotsPerXcc = 0;
    SlotMap* maps = nullptr;
    std::shared_mutex gate;
    std::mutex m; int lanes = 0;
};

static bool g_pageableStaging = false;

// A launch whose largest coder table does not fit an arena slot -- the <8,6> model of 8-bin quality scores and the full <256,1>
// model are 32 MiB, a slot 16.8 MB -- runs ALONE on the device (encode_streams_raw: exclusive), one slice after the other: a
// --reduced library's step was 7.1 s that way against 1.15 s with its fourteen slices side by side (profiles/r03_reduced_mode.txt).
// So the first such launch makes the pool's slots big enough, once, with nothing in flight (the gate taken alone): same number of
// slots, 3 072 x 32 MiB = 103 GB of the 288.  Only then: lossless packs -- and the helper pipelines of libraries of several
// batches, each with a pool of its own -- keep the 53 GB pool.  No room for it: the exclusive launches stay.
static void pool_grow(Device* dev, Pool* pool, uint64_t need)
{
    const uint64_t big = ((need + kGuard) + 4095ull) & ~4095ull;
    std::unique_lock<std::shared_mutex> alone(pool->gate);                    // every launch in flight has drained
    if (big <= pool->slotStride) return;                                    // another lane was first
    const uint64_t want = (uint64_t)pool->slotsPerXcc * kXcc * big;
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) != hipSuccess || (uint64_t)freeB + pool->bytes < want + (24ull << 30)) return;      // (24 GB stay for the batches' buffers)
    uint8_t* old = pool->arenas;
    if (hipFree(old) != hipSuccess) return;
    uint8_t* fresh = nullptr;
    if (hipMalloc((void**)&fresh, want) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc((void**)&fresh, pool->bytes) != hipSuccess) { pool->arenas = nullptr; snprintf(dev->err, sizeof dev->err, "arena pool lost while growing it"); return; }
        pool->arenas = fresh; return;
    }
    pool->arenas = fresh; pool->bytes = want; pool->slotStride = big;
    if (getenv("FS_TRACE")) fprintf(stderr, "[trace] arena pool grown to %u slots of %.1f MB (%.1f GB) for a %.1f MB coder table\n", pool->slotsPerXcc * kXcc, big / 1e6, want / 1e6 / 1e3, need / 1e6);
}

static int lane_init(Device* dev, char* err, size_t errLen)
{
    hipError_t e;
    auto fail = [&](const char* what, hipError_t c) { snprintf(err, errLen, "%s: %s", what, hipGetErrorString(c)); return -1; };
    if ((e = hipSetDevice(dev->deviceId)) != hipSuccess) return fail("hipSetDevice", e);
    dev->stagePageable = g_pageableStaging ? 1u : 0u;
    dev->trace = getenv("FS_TRACE") ? 1u : 0u;
    if ((e = hipStreamCreateWithFlags((hipStream_t*)&dev->stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    if ((e = hipMalloc((void**)&dev->queueHead, 64)) != hipSuccess) return fail("hipMalloc(queue)", e);
    for (int i = 0; i < 6; ++i) if ((e = hipEventCreate((hipEvent_t*)&dev->ev[i])) != hipSuccess) return fail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags((hipEvent_t*)&dev->evWait, hipEventBlockingSync | hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
    return 0;
}

int device_create(Device** out, int deviceId, uint32_t maxWaves, char* err, size_t errLen)
{
    want_hw_queues();
    const double tc0 = wallMs();
    Device* dev = new Device();
    memset(dev, 0, sizeof *dev);
    *out = nullptr;
    Pool* pool = new Pool();
    auto fail = [&](const char* what, hipError_t e) {
        snprintf(err, errLen, "%s: %s", what, hipGetErrorString(e));
        if (pool->arenas) (void)hipFree(pool->arenas);
        if (pool->maps) (void)hipFree(pool->maps);
        delete pool; delete dev; return -1;
    };
    hipError_t e;
    const bool trace = getenv("FS_TRACE") != nullptr;
    double tc1 = tc0;
    auto lap = [&](const char* what) { if (trace) { const double t = wallMs(); fprintf(stderr, "[trace] device_create: %s %.1f ms\n", what, t - tc1); tc1 = t; } };
    if ((e = hipSetDevice(deviceId)) != hipSuccess) return fail("hipSetDevice", e);
    lap("runtime start-up (hipSetDevice)");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, deviceId)) != hipSuccess) return fail("hipGetDeviceProperties", e);
    dev->deviceId = deviceId;
    dev->cus = prop.multiProcessorCount;
    snprintf(dev->name, sizeof dev->name, "%s", prop.gcnArchName);
    size_t freeB = 0, totalB = 0;
    if ((e = hipMemGetInfo(&freeB, &totalB)) != hipSuccess) return fail("hipMemGetInfo", e);
    // One arena slot per resident wavefront (kWavesPerSimd x 4 SIMDs per CU), in kXcc equal partitions; the pool takes
    // at most 55 % of the free HBM.
    const uint32_t waves = maxWaves ? maxWaves : (uint32_t)dev->cus * 4u * kWavesPerSimd;
    // (a slot holds a PPMd arena, 16.8 MB; the pool grows to slots of a coder table's size when a launch first needs one: pool_grow)
    const uint64_t stride = ((fsppmd::ARENA_BYTES + kGuard) + 4095ull) & ~4095ull;
    uint32_t perXcc = std::min<uint32_t>((waves + kXcc - 1) / kXcc, kBitmapWords * 64u);
    const uint64_t budget = (uint64_t)((double)freeB * 0.55);
    while (perXcc > 1 && (uint64_t)perXcc * kXcc * stride > budget) --perXcc;
    pool->slotStride = stride; pool->slotsPerXcc = perXcc; pool->bytes = (uint64_t)perXcc * kXcc * stride;
    if (pool->bytes > budget) { snprintf(err, errLen, "not enough device memory for the coder arenas"); delete pool; delete dev; return -1; }
    dev->nWaves = waves; dev->pool = pool;
    lap("properties, memory info");
    if ((e = hipMalloc((void**)&pool->arenas, pool->bytes)) != hipSuccess) return fail("hipMalloc(arenas)", e);
    lap("hipMalloc of the arena pool");
    if ((e = hipMalloc((void**)&pool->maps, sizeof(SlotMap) * kXcc)) != hipSuccess) return fail("hipMalloc(slot maps)", e);
    if ((e = hipMemset(pool->maps, 0, sizeof(SlotMap) * kXcc)) != hipSuccess) return fail("hipMemset(slot maps)", e);
    lap("slot maps (first memset: code objects)");
    if (lane_init(dev, err, errLen) != 0) { (void)hipFree(pool->arenas); (void)hipFree(pool->maps); delete pool; delete dev; return -1; }
    pool->lanes = 1;
    *out = dev;
    lap("first lane (stream, events)");
    if (trace) fprintf(stderr, "[trace] device_create: up to %u waves, %u slots per XCD, %.1f GB arena pool, %.1f ms\n", waves, perXcc, pool->bytes / 1e9, wallMs() - tc0);
    return 0;
}

int lane_create(Device* first, Device** out, char* err, size_t errLen)
{
    Device* dev = new Device();
    memset(dev, 0, sizeof *dev);
    *out = nullptr;
    dev->deviceId = first->deviceId; dev->cus = first->cus; dev->nWaves = first->nWaves; dev->pool = first->pool;
    memcpy(dev->name, first->name, sizeof dev->name);
    if (lane_init(dev, err, errLen) != 0) { delete dev; return -1; }
    { std::lock_guard<std::mutex> g(dev->pool->m); dev->pool->lanes++; }
    *out = dev;
    return 0;
}

void device_destroy(Device* dev)
{
    if (!dev) return;
    (void)hipSetDevice(dev->deviceId);
    if (dev->stream) (void)hipStreamSynchronize((hipStream_t)dev->stream);
    const double td0 = wallMs();
    void* ptrs[] = {dev->queueHead, dev->dIn, dev->dScratch, dev->dItems, dev->dOrder, dev->dSizes, dev->dRestarts, dev->dPlans, dev->dBlocks};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (uint32_t i = 0; i < dev->nOldDev; ++i) (void)hipFree(dev->oldDev[i]);
    const double td1 = wallMs();
    if (dev->evWait) (void)hipEventDestroy((hipEvent_t)dev->evWait);
    if (dev->hStage) pinned_free(dev->hStage, dev->capStage, !dev->stagePageable);
    for (uint32_t i = 0; i < dev->nOldStage; ++i) pinned_free(dev->oldStage[i], dev->oldStageCap[i], !dev->stagePageable);
    const double td2 = wallMs();
    for (int i = 0; i < 6; ++i) if (dev->ev[i]) (void)hipEventDestroy((hipEvent_t)dev->ev[i]);
    if (dev->stream) (void)hipStreamDestroy((hipStream_t)dev->stream);
    if (dev->trace) fprintf(stderr, "[trace] lane teardown: device buffers %.1f ms, staging buffer %.1f ms, events + stream %.1f ms\n", td1 - td0, td2 - td1, wallMs() - td2);
    if (Pool* pool = dev->pool) {
        bool last;
        { std::lock_guard<std::mutex> g(pool->m); last = --pool->lanes == 0; }
        if (last) {
            const double t0 = wallMs();
            (void)hipFree(pool->arenas); (void)hipFree(pool->maps); delete pool;
            if (getenv("FS_TRACE")) fprintf(stderr, "[trace] device_destroy: hipFree of the arena pool %.1f ms\n", wallMs() - t0);
        }
    }
    delete dev;
}

int lane_debug(Device* dev, char* out, size_t outLen)
{
    if (!dev || !out || outLen == 0) return -1;
    out[0] = 0;
    if (hipSetDevice(dev->deviceId) != hipSuccess) return -1;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return -1;
    uint32_t head = 0; SlotMap maps[kXcc]; memset(maps, 0, sizeof maps);
    bool ok = hipMemcpyAsync(&head, dev->queueHead, 4, hipMemcpyDeviceToHost, s) == hipSuccess;
    ok = ok && hipMemcpyAsync(maps, dev->pool->maps, sizeof maps, hipMemcpyDeviceToHost, s) == hipSuccess;
    ok = ok && hipStreamSynchronize(s) == hipSuccess;
    (void)hipStreamDestroy(s);
    if (!ok) { snprintf(out, outLen, "device read failed"); return -1; }
    int n = snprintf(out, outLen, "queue head %u, stream %s; arena slots taken per XCD:", head, hipStreamQuery((hipStream_t)dev->stream) == hipSuccess ? "idle" : "busy");
    for (uint32_t x = 0; x < kXcc && n > 0 && (size_t)n < outLen; ++x) { int c = 0; for (uint32_t w = 0; w < kBitmapWords; ++w) c += __builtin_popcountll(maps[x].w[w]); n += snprintf(out + n, outLen - n, " %d", c); }
    return 0;
}

void set_pageable_staging(bool on) { g_pageableStaging = on; }
bool pageable_staging() { return g_pageableStaging; }

void* pinned_alloc(size_t* bytes, bool pin)
{
    const size_t len = (*bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    void* p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return nullptr;
    (void)madvise(p, len, MADV_HUGEPAGE);
    const long pg = sysconf(_SC_PAGESIZE);
    for (size_t o = 0; o < len; o += (size_t)pg) ((volatile uint8_t*)p)[o] = 0;
    if (pin && hipHostRegister(p, len, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); munmap(p, len); return nullptr; }
    *bytes = len;
    return p;
}

void pinned_free(void* p, size_t bytes, bool pinned)
{
    if (!p) return;
    if (pinned) (void)hipHostUnregister(p);
    munmap(p, bytes);
}

uint8_t* staging_buffer(Device* dev, size_t bytes)
{
    if (bytes <= dev->capStage && dev->hStage) return dev->hStage;
    (void)hipSetDevice(dev->deviceId);                   // lane threads start on device 0
    if (dev->hStage) {
        if (dev->nOldStage < 4u) { dev->oldStage[dev->nOldStage] = dev->hStage; dev->oldStageCap[dev->nOldStage] = dev->capStage; ++dev->nOldStage; }
        else pinned_free(dev->hStage, dev->capStage, !dev->stagePageable);
    }
    dev->hStage = nullptr; dev->capStage = 0;
    size_t want = bytes + bytes / 4 + 4096;
    dev->hStage = (uint8_t*)pinned_alloc(&want, !dev->stagePageable);
    if (!dev->hStage) { snprintf(dev->err, sizeof dev->err, "pinned staging buffer of %zu bytes failed", want); return nullptr; }
    dev->capStage = want;
    return dev->hStage;
}

void staging_release(Device* dev)
{
    if (dev->hStage) pinned_free(dev->hStage, dev->capStage, !dev->stagePageable);
    dev->hStage = nullptr; dev->capStage = 0;
    for (uint32_t i = 0; i < dev->nOldStage; ++i) pinned_free(dev->oldStage[i], dev->oldStageCap[i], !dev->stagePageable);
    dev->nOldStage = 0;
}

// H2D + fs_encode_streams + D2H of the per-stream sizes.  Leaves the coded streams in dev->dScratch.
// host side of the two kernels: the chunk table of the plan's jobs (kept by the caller until the stream has been waited for), the
// scratch behind `scratchBase` in the lane's input buffer (chunk table, chunk totals, per-read counts), the two launches
struct IdLaunch { std::vector<IdChunk> chunks; uint64_t tableOff = 0, totalsOff = 0, countsOff = 0, bytes = 0; };
static void id_launch_plan(const IdJob* jb, uint32_t nJobs, uint32_t nStrings, uint64_t scratchBase, IdLaunch& L)
{
    L.chunks.clear();
    for (uint32_t j = 0; j < nJobs; ++j) {
        const uint32_t first = (uint32_t)L.chunks.size();
        uint32_t c0 = 0;
        do { L.chunks.push_back(IdChunk{j, c0, first, 0u}); c0 += 256u; } while (c0 < jb[j].count);      // (an empty bin keeps one chunk: it sets the lengths to zero)
    }
    L.tableOff = (scratchBase + 15u) & ~15ull;
    L.totalsOff = L.tableOff + sizeof(IdChunk) * L.chunks.size();
    L.countsOff = L.totalsOff + 8ull * L.chunks.size();
    L.bytes = L.countsOff + 4ull * nStrings + 16u - scratchBase;
}
static int id_launch(Device* dev, hipStream_t st, const IdPlan& plan, uint64_t gatherBase, const IdLaunch& L)
{
    if (L.chunks.empty()) return 0;
    HIP_TRY(hipMemcpyAsync(dev->dIn + L.tableOff, L.chunks.data(), sizeof(IdChunk) * L.chunks.size(), hipMemcpyHostToDevice, st));
    const IdChunk* ck = (const IdChunk*)(dev->dIn + L.tableOff);
    const IdJob* jobs = (const IdJob*)(dev->dIn + plan.jobs_off); const IdString* strs = (const IdString*)(dev->dIn + plan.strings_off);
    uint32_t* counts = (uint32_t*)(dev->dIn + L.countsOff); uint32_t* totals = (uint32_t*)(dev->dIn + L.totalsOff);
    hipLaunchKernelGGL(fs_id_count, dim3((uint32_t)L.chunks.size()), dim3(256), 0, st, ck, jobs, strs, (const uint8_t*)dev->dIn, counts, totals);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(fs_id_write, dim3((uint32_t)L.chunks.size()), dim3(256), 0, st, ck, jobs, strs, (const uint8_t*)dev->dIn, (const uint32_t*)counts, (const uint32_t*)totals,
                       (uint8_t*)(dev->dIn + gatherBase), (StreamItem*)dev->dItems);
    HIP_TRY(hipGetLastError());
    return 0;
}


// the emission's scratch behind the streams it writes: per op its two counts and its two places, per bin its channels' totals
struct EmitLaunch { uint64_t countsOff = 0, offsOff = 0, totalsOff = 0, bytes = 0; };
static void emit_launch_plan(const EmitPlan& plan, uint64_t scratchBase, EmitLaunch& L)
{
    L.countsOff = (scratchBase + 15u) & ~15ull;
    L.offsOff = L.countsOff + 8ull * plan.n_ops;
    L.totalsOff = L.offsOff + 8ull * plan.n_ops;
    L.bytes = L.totalsOff + 4ull * (ECH_COUNT + 1u) * plan.n_jobs + 16u - scratchBase;
}
// (every op inside its bin's bases and contig bytes, every stream with room for the most its ops can write: fsemit::plan_error,
// emit_core.h -- checked on the host, the kernels trust their descriptors)
static int emit_check(Device* dev, const uint8_t* input, size_t inputBytes, const EmitPlan& plan, uint32_t nItems)
{
    uint32_t badJob = 0;
    const char* why = fsemit::plan_error(input, inputBytes, plan, nItems, badJob);
    if (why) { snprintf(dev->err, sizeof dev->err, "emission job %u: %s", badJob, why); return -1; }
    return 0;
}
static int emit_launch(Device* dev, hipStream_t st, const EmitPlan& plan, uint64_t emitBase, const EmitLaunch& L)
{
    if (plan.n_jobs == 0) return 0;
    const EmitJob* jobs = (const EmitJob*)(dev->dIn + plan.jobs_off); const EmitOp* ops = (const EmitOp*)(dev->dIn + plan.ops_off);
    uint32_t* counts = (uint32_t*)(dev->dIn + L.countsOff); uint32_t* offs = (uint32_t*)(dev->dIn + L.offsOff); uint32_t* totals = (uint32_t*)(dev->dIn + L.totalsOff);
    uint8_t* out = (uint8_t*)(dev->dIn + emitBase);
    if (plan.n_ops) {
        hipLaunchKernelGGL(fs_emit_count, dim3((plan.n_ops + 3u) / 4u), dim3(256), 0, st, jobs, ops, plan.n_ops, (const uint8_t*)dev->dIn, counts);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(fs_emit_scan, dim3(plan.n_jobs), dim3(256), 0, st, jobs, ops, (const uint32_t*)counts, offs, totals, (StreamItem*)dev->dItems, out);
    HIP_TRY(hipGetLastError());
    if (plan.n_ops) {
        hipLaunchKernelGGL(fs_emit_write, dim3((plan.n_ops + 3u) / 4u), dim3(256), 0, st, jobs, ops, plan.n_ops, (const uint8_t*)dev->dIn, (const uint32_t*)offs, out);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(fs_rle_binary, dim3(3u * plan.n_jobs), dim3(256), 0, st, jobs, (const uint32_t*)totals, out, (StreamItem*)dev->dItems);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(fs_rle0, dim3(plan.n_jobs), dim3(256), 0, st, jobs, (const uint32_t*)(dev->dIn + plan.ids_off), out, (StreamItem*)dev->dItems);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int run_encode(Device* dev, const uint8_t* input, size_t inputBytes, std::vector<StreamItem>& items,
                      std::vector<uint32_t>& sizes, uint64_t& scratchBytes, BatchTiming* timing, const GatherPlan* gather = nullptr, const IdPlan* ids = nullptr, const EmitPlan* emit = nullptr)
{
    HIP_TRY(hipSetDevice(dev->deviceId));
    hipStream_t st = (hipStream_t)dev->stream;
    const uint32_t nItems = (uint32_t)items.size();
    uint64_t scratch = 0;
    for (auto& it : items) { it.out_off = scratch; scratch += ((uint64_t)it.out_cap + 15u) & ~15ull; }
    scratchBytes = scratch;
    // a QVZ stream adapts a private copy of its library's statistics image (model blob header in the input buffer)
    auto qvzImageBytes = [&](const StreamItem& s) -> uint64_t {
        fsqvz::ModelHeader h; memcpy(&h, input + s.aux_off, sizeof h);
        return 4ull * h.image_words;
    };
    for (const auto& it : items)
        if (it.kind == KIND_QVZ) {
            if ((it.aux_off & 15u) || it.aux_off + sizeof(fsqvz::ModelHeader) > inputBytes) { snprintf(dev->err, sizeof dev->err, "QVZ stream item without a model blob"); return -1; }
            fsqvz::ModelHeader h; memcpy(&h, input + it.aux_off, sizeof h);
            if (it.aux_off + fsqvz::blob_bytes(h.n_ctx, h.image_words) > inputBytes || (it.in_off & 3u)) { snprintf(dev->err, sizeof dev->err, "QVZ model blob outside the batch input"); return -1; }
        }
    // longest-first queue order (a PPMd symbol costs roughly 10x a range-coder symbol)
    std::vector<uint32_t> order(nItems);
    for (uint32_t i = 0; i < nItems; ++i) order[i] = i;
    auto cost = [&](uint32_t i) -> uint64_t {
        const StreamItem& s = items[i];
        if (s.kind == KIND_PPMD) return (uint64_t)s.in_len * 10u + 2000u;
        if (s.kind == KIND_QVZ) return (uint64_t)s.in_len * 4u + qvzImageBytes(s) / 256u + 100u;
        const uint64_t tbl = fsrc::model_table_bytes(s.kind - KIND_RC_BASE);
        return (uint64_t)s.in_len * 2u + tbl / 512u + 100u;
    };
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost(a) > cost(b); });

    // the gathered quality streams live behind the uploaded bytes
    // (quality streams first, then the read-id streams)
    // (quality streams first, then the read-id streams, then what the emission kernels write)
    const uint64_t emitBase = (((uint64_t)inputBytes + 15u) & ~15ull) + (gather ? gather->out_bytes : 0) + (ids ? ids->out_bytes : 0);
    const uint64_t gatherBase = ((uint64_t)inputBytes + 15u) & ~15ull, gatherBytes = (gather ? gather->out_bytes : 0) + (ids ? ids->out_bytes : 0) + (emit ? emit->out_bytes : 0);
    if (emit && emit->n_jobs && emit_check(dev, input, inputBytes, *emit, (uint32_t)items.size())) return -1;
    if (ids && (ids->jobs_off & 7u || ids->strings_off & 7u || ids->jobs_off + (uint64_t)ids->n_jobs * sizeof(IdJob) > inputBytes || ids->strings_off + (uint64_t)ids->n_strings * sizeof(IdString) > inputBytes)) {
        snprintf(dev->err, sizeof dev->err, "read-id plan outside the batch input"); return -1;
    }
    const size_t descBytes = gather ? (gather->qvz ? sizeof(QuaQvzString) : (gather->bits == 6u ? sizeof(QuaString) : sizeof(QuaPairString))) : 0;
    if (gather && (gather->desc_off & 15u || gather->desc_off + (uint64_t)gather->n_strings * descBytes > inputBytes || gatherBase + gatherBytes > 0xFFFFFF00ull ||
                   (gather->bits != 6u && gather->bits != 3u && gather->bits != 1u) || (gather->bits != 6u && gather->n_list_off + gather->n_list_bytes > inputBytes))) {
        snprintf(dev->err, sizeof dev->err, "quality gather plan outside the batch input"); return -1;
    }
    IdLaunch idl; EmitLaunch eml;
    if (ids && ids->n_jobs) id_launch_plan((const IdJob*)(input + ids->jobs_off), ids->n_jobs, ids->n_strings, gatherBase + gatherBytes, idl);
    if (emit && emit->n_jobs) emit_launch_plan(*emit, gatherBase + gatherBytes + idl.bytes, eml);
    if (gatherBase + gatherBytes + idl.bytes + eml.bytes > 0xFFFFFF00ull) { snprintf(dev->err, sizeof dev->err, "batch input and device-written streams beyond 4 GiB"); return -1; }
    if (ensure(dev, dev->dIn, dev->capIn, gatherBase + gatherBytes + idl.bytes + eml.bytes + kInSlack)) return -1;
    if (ensure(dev, dev->dScratch, dev->capScratch, scratch + 16)) return -1;
    if (ensure(dev, dev->dItems, dev->capItems, sizeof(StreamItem) * nItems)) return -1;
    if (ensure(dev, dev->dOrder, dev->capOrder, 4ull * nItems)) return -1;
    if (ensure(dev, dev->dSizes, dev->capSizes, 4ull * nItems)) return -1;
    if (ensure(dev, dev->dRestarts, dev->capRestarts, 64ull * nItems)) return -1;

    uint64_t need = fsppmd::ARENA_BYTES;
    for (const auto& it : items) {
        if (it.kind == KIND_QVZ) need = std::max<uint64_t>(need, qvzImageBytes(it));
        else if (it.kind != KIND_PPMD) need = std::max<uint64_t>(need, fsrc::model_table_bytes(it.kind - KIND_RC_BASE));
    }
    Pool* pool = dev->pool;
    // The gate is held from here until this launch has drained: the slot size is read under it (every launch in flight at
    // the same time must place its slots by the same stride; pool_grow changes it only with the gate taken alone), and
    // exclusive launches never overlap ring launches.
    std::shared_lock<std::shared_mutex> shared(pool->gate);
    if (((need + kGuard + 4095ull) & ~4095ull) > pool->slotStride) { shared.unlock(); pool_grow(dev, pool, need); shared.lock(); }
    if (!pool->arenas) return -1;
    const bool exclusive = ((need + kGuard + 4095ull) & ~4095ull) > pool->slotStride;      // table larger than an arena slot (and the pool could not grow)
    std::unique_lock<std::shared_mutex> alone(pool->gate, std::defer_lock);
    if (exclusive) { shared.unlock(); alone.lock(); }
    const uint64_t stride = exclusive ? ((need + kGuard) + 4095ull) & ~4095ull : pool->slotStride;
    uint32_t maxLen = 0;
    for (const auto& it : items) if (it.kind == KIND_PPMD) maxLen = std::max(maxLen, it.in_len);
    uint32_t maxAny = maxLen;                                                 // (... or QVZ stream, or stream of a range coder with a windowed form: what a --lossy / --reduced launch ends with)
    for (const auto& it : items) if (it.kind == KIND_QVZ || (it.kind != KIND_PPMD && it.kind - KIND_RC_BASE <= fsrc::M_A8O6)) maxAny = std::max(maxAny, it.in_len);
    const uint32_t longLen = std::max(1u, maxAny / 2);                     // "long" = at least half of the longest such stream
    const uint32_t nRest = nItems;
    HIP_TRY(hipMemcpyAsync(dev->dIn, input, inputBytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dev->dItems, items.data(), sizeof(StreamItem) * nItems, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dev->dOrder, order.data(), 4ull * nRest, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(dev->queueHead, 0, 64, st));
    HIP_TRY(hipMemsetAsync(dev->dSizes, 0xFF, 4ull * nItems, st));          // a size no wave writes reads as 0xFFFFFFFF: the callers' error path, not last launch's value
    if (gather && gather->n_strings) {
        // every string's source and destination inside the buffer, checked here: the kernel trusts its descriptors
        if (gather->qvz) {
            // ... and the tables it reads through them: the library's header, every table and the generator's outputs inside the input
            if (gather->bits != 6u) { snprintf(dev->err, sizeof dev->err, "QVZ quality gather of other than six-bit scores"); return -1; }
            const QuaQvzString* qs = (const QuaQvzString*)(input + gather->desc_off);
            uint64_t lastModel = ~0ull;
            for (uint32_t i = 0; i < gather->n_strings; ++i) {
                const uint64_t mo = 16ull * qs[i].model16;
                if (mo != lastModel) {
                    QvzSymHeader h;
                    bool ok = mo + sizeof h <= inputBytes;
                    if (ok) {
                        memcpy(&h, input + mo, sizeof h);
                        const uint64_t end = h.total_bytes;
                        ok = mo + end <= inputBytes && h.columns >= 1u && h.columns <= 255u && (h.n_ctx & 1u) == 0u && h.n_ctx < (1u << 24) &&
                             (h.col_ctx_base_off & 3u) == 0u && h.col_ctx_base_off + 4ull * h.columns <= end && (h.col_index_off & 1u) == 0u && h.col_index_off + 164ull * h.columns <= end &&
                             h.qratio_off + (uint64_t)h.n_ctx / 2u <= end && h.quant_off + 72ull * h.n_ctx <= end && h.state_of_off + 72ull * h.n_ctx <= end &&
                             (h.well_off & 3u) == 0u && h.well_off + 4ull * h.well_words <= end;
                        // (a column's first context + an input-alphabet index stays inside the table: the kernel checks ctx < n_ctx itself)
                    }
                    if (!ok) { snprintf(dev->err, sizeof dev->err, "QVZ quantizer tables outside the batch input"); return -1; }
                    lastModel = mo;
                }
                if ((qs[i].src_bit >> 3) + (6ull * qs[i].len + 7u) / 8u + 2u > inputBytes || (qs[i].dst_off & 3u) || (uint64_t)qs[i].dst_off + 4ull * qs[i].len > gatherBytes) {
                    snprintf(dev->err, sizeof dev->err, "quality string %u outside the batch input", i); return -1;
                }
            }
        } else if (gather->bits == 6u) {
            const QuaString* qs = (const QuaString*)(input + gather->desc_off);
            for (uint32_t i = 0; i < gather->n_strings; ++i)
                if ((qs[i].src_bit >> 3) + (6ull * qs[i].len + 7u) / 8u + 4u > inputBytes || (uint64_t)qs[i].dst_off + qs[i].len > gatherBytes) {
                    snprintf(dev->err, sizeof dev->err, "quality string %u outside the batch input", i); return -1;
                }
        } else {
            const QuaPairString* qs = (const QuaPairString*)(input + gather->desc_off);
            for (uint32_t i = 0; i < gather->n_strings; ++i)
                if ((qs[i].src_bit >> 3) + ((uint64_t)gather->bits * qs[i].len + 7u) / 8u + 2u > inputBytes || qs[i].n_count > qs[i].len ||
                    2ull * ((uint64_t)qs[i].dst_off + qs[i].len - qs[i].n_count) > gatherBytes || (uint64_t)qs[i].n_off + qs[i].n_count > gather->n_list_bytes) {
                    snprintf(dev->err, sizeof dev->err, "quality string %u outside the batch input", i); return -1;
                }
        }
        HIP_TRY(hipEventRecord((hipEvent_t)dev->ev[4], st));
        const uint32_t blocks = std::min<uint32_t>((gather->n_strings + 3u) / 4u, (uint32_t)dev->cus * 16u);
        if (gather->qvz)      // a read per lane
            hipLaunchKernelGGL(fs_gather_quality_qvz, dim3(std::min<uint32_t>((gather->n_strings + 255u) / 256u, (uint32_t)dev->cus * 32u)), dim3(256), 0, st,
                               (const QuaQvzString*)(dev->dIn + gather->desc_off), gather->n_strings, (const uint8_t*)dev->dIn, (uint8_t*)(dev->dIn + gatherBase));
        else if (gather->bits == 6u)
            hipLaunchKernelGGL(fs_gather_quality, dim3(blocks), dim3(256), 0, st, (const QuaString*)(dev->dIn + gather->desc_off), gather->n_strings,
                               (const uint8_t*)dev->dIn, (uint8_t*)(dev->dIn + gatherBase));
        else
            hipLaunchKernelGGL(fs_gather_quality_pairs, dim3(blocks), dim3(256), 0, st, (const QuaPairString*)(dev->dIn + gather->desc_off), gather->n_strings,
                               (const uint8_t*)dev->dIn, (const uint8_t*)(dev->dIn + gather->n_list_off), (uint8_t*)(dev->dIn + gatherBase), gather->bits, gather->sym_of_bit[0], gather->sym_of_bit[1]);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord((hipEvent_t)dev->ev[5], st));
    }
    const uint32_t grid = exclusive ? (uint32_t)std::min<uint64_t>(std::min<uint64_t>(nItems, dev->nWaves), pool->bytes / stride)
                                    : std::min<uint32_t>(std::max(nRest, 1u), dev->nWaves);
    // (Round 4 tried to keep the short streams from crowding the long ones out of the L2s -- a launch with a workgroup per stream of a
    // million symbols and more and only k more for everything else: 2 305 / 2 382 MB/s without, 2 283 with k = 256, 1 963 / 1 974 with 64,
    // 831 with 16, profiles/r04_short_waves_sweep.txt: the short streams need the resident waves more than the long ones need the cache.)
    if (grid == 0) { snprintf(dev->err, sizeof dev->err, "arena pool too small for a %llu-byte coder table", (unsigned long long)need); return -1; }
    if (ids && ids->n_jobs) {
        // every job's strings, table and output inside the buffers, checked here: the kernel trusts its descriptors (and may read up
        // to seven bytes past a string's end: kInSlack)
        const IdJob* jb = (const IdJob*)(input + ids->jobs_off); const IdString* ss = (const IdString*)(input + ids->strings_off);
        for (uint32_t j = 0; j < ids->n_jobs; ++j) {
            bool ok = (uint64_t)jb[j].first + jb[j].count <= ids->n_strings && jb[j].tok_item < nItems && jb[j].val_item < nItems && jb[j].table_off + 8u <= inputBytes && (jb[j].table_off & 7u) == 0 &&
                      jb[j].tok_out + 2ull * items[jb[j].tok_item].in_len <= gatherBytes && jb[j].val_out + 2ull * items[jb[j].val_item].in_len <= gatherBytes;
            for (uint32_t k = 0; ok && k < jb[j].count; ++k) { const IdString& s = ss[jb[j].first + k]; ok = s.len <= 255u && id_string_inside(s.src_bit, s.len, inputBytes); }
            if (!ok) { snprintf(dev->err, sizeof dev->err, "read-id job %u outside the batch input", j); return -1; }
        }
        if (id_launch(dev, st, *ids, gatherBase, idl)) return -1;
    }
    if (emit && emit->n_jobs && emit_launch(dev, st, *emit, emitBase, eml)) return -1;
    HIP_TRY(hipEventRecord((hipEvent_t)dev->ev[0], st));
    std::vector<uint32_t> restarts, writtenLen;
    uint32_t qHead = 0, tailLaunches = 0;
    {
        EncodeArgs ka;
        ka.items = (const StreamItem*)dev->dItems; ka.order = (const uint32_t*)dev->dOrder; ka.in = (const uint8_t*)dev->dIn; ka.out = (uint8_t*)dev->dScratch;
        ka.outSizes = (uint32_t*)dev->dSizes; ka.restarts = (uint32_t*)dev->dRestarts; ka.arenas = pool->arenas; ka.arenaStride = stride;
        ka.queueHead = (uint32_t*)dev->queueHead; ka.maps = exclusive ? (SlotMap*)nullptr : pool->maps;
        ka.nItems = nRest; ka.longLen = longLen; ka.slotsPerXcc = pool->slotsPerXcc;
        { const char* wb = getenv("FS_WG_BUDGET"); ka.budget = exclusive ? 0u : (wb ? (uint32_t)atoll(wb) : (1u << 20)); }      // (the variable, read per launch: A/B runs and the tests; 0 = no bound)
        // two-wave form where the step is bound by its longest PPMd stream (the coder runs beside the model walk: ~1.4x per
        // stream, but a stream takes two wave slots)
        // (FS_WAVES=1/2 forces a form: the tests run every stream through both.  Round 3's three-wave form -- windows prepared by a
        // wave of their own ahead of the serial walk -- was bit-exact and slower, 1.10 s against 0.97 s on a lone 7 M-symbol
        // stream, and left the tree in round 4; its measurements stay in profiles/r03_three_wave_*.txt.)
        uint32_t waves = maxLen >= (256u << 10) ? 2u : 1u;
        if (const char* tw = getenv("FS_WAVES")) waves = (uint32_t)std::max(1, std::min(2, atoi(tw)));
        // Launches whose range-coded symbols (small alphabets: the models with a windowed form) weigh beside their PPMd symbols
        // -- the quality scores of a --reduced or --max library -- take the kernels with the windowed coders.  A lossless
        // launch (flags and letters: a few per cent of its PPMd symbols) keeps the kernels it had: with a plain length
        // threshold every launch of the BASELINE library qualified, and its step was 1-2 % longer for it.
        // (a --lossy library's QVZ streams likewise: the same kernels put their fractions and the interval's pass on the coder wave)
        // (The streams the emission kernels write -- match bits, letters -- do not count: what the host knows of them is the ROOM they were
        // given, a multiple of their size, and with it every lossless launch took these kernels: found in the round's kernel statistics.)
        std::vector<uint8_t> roomOnly(items.size(), 0);
        if (emit && emit->n_jobs) {
            const EmitJob* ej = (const EmitJob*)(input + emit->jobs_off);
            for (uint32_t j = 0; j < emit->n_jobs; ++j) for (uint32_t ch = 0; ch <= ECH_COUNT; ++ch) if (ej[j].item[ch] < items.size()) roomOnly[ej[j].item[ch]] = 1;
        }
        uint64_t sumRc = 0, sumPpmd = 0;
        for (size_t i = 0; i < items.size(); ++i) {
            const auto& it = items[i];
            if (it.kind == KIND_PPMD) sumPpmd += it.in_len;
            else if (!roomOnly[i] && (it.kind == KIND_QVZ || it.kind - KIND_RC_BASE <= fsrc::M_A8O6)) sumRc += it.in_len;
        }
        bool rcWin = sumRc >= 4096u && 4u * sumRc >= sumPpmd;
        {   // the coder wave takes the range coder's pass: worth a second wave per stream where a long range-coded stream ends the launch
            uint32_t maxRc = 0;
            for (size_t i = 0; i < items.size(); ++i) { const auto& it = items[i]; if (!roomOnly[i] && it.kind != KIND_PPMD && (it.kind == KIND_QVZ || it.kind - KIND_RC_BASE <= fsrc::M_A8O6)) maxRc = std::max(maxRc, it.in_len); }
            if (rcWin && maxRc >= (256u << 10) && !getenv("FS_WAVES")) waves = 2u;
        }
        if (const char* rw = getenv("FS_RC_WINDOWS")) rcWin = atoi(rw) != 0;
        auto launch = [&]() -> hipError_t {
            if (waves == 2u) { const uint32_t g2 = std::max(1u, std::min(grid, dev->nWaves / 2u)); if (rcWin) hipLaunchKernelGGL(fs_encode_streams2_w, dim3(g2), dim3(128), 0, st, ka); else hipLaunchKernelGGL(fs_encode_streams2, dim3(g2), dim3(128), 0, st, ka); }
            else if (rcWin) hipLaunchKernelGGL(fs_encode_streams_w, dim3(grid), dim3(64), 0, st, ka);
            else hipLaunchKernelGGL(fs_encode_streams, dim3(grid), dim3(64), 0, st, ka);
            return hipGetLastError();
        };
        HIP_TRY(launch());
        HIP_TRY(hipEventRecord((hipEvent_t)dev->ev[1], st));
        if (dev->trace) { HIP_TRY(wait_stream(dev, st)); }
        sizes.resize(nItems);
        restarts.resize(16ull * nItems);
        HIP_TRY(hipMemcpyAsync(&qHead, dev->queueHead, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(sizes.data(), dev->dSizes, 4ull * nItems, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(restarts.data(), dev->dRestarts, 64ull * nItems, hipMemcpyDeviceToHost, st));
        // (the streams the device's own kernels wrote: their lengths, for the statistics -- the host knows only the room it gave them)
        if (emit && emit->n_jobs) { writtenLen.resize(nItems); HIP_TRY(hipMemcpy2DAsync(writtenLen.data(), 4, (const uint8_t*)dev->dItems + offsetof(StreamItem, in_len), sizeof(StreamItem), 4, nItems, hipMemcpyDeviceToHost, st)); }
        HIP_TRY(wait_stream(dev, st));
        // The tail launch.  A workgroup that finds no free arena slot on its XCD leaves at once (no workgroup ever waits for another:
        // encode_streams_body), and the launch's other workgroups take its streams; only when NONE of a launch's workgroups found a slot
        // -- or those that did left before the queue was... they never do: they leave when it is empty -- is the queue's head short of
        // its end here.  Then the launch is made again (same queue, same arguments) until every stream has been taken; whoever takes a
        // stream codes it to its end.  The slots are held by this pool's other launches, which drain on their own, so every pass ends.
        uint32_t idle = 0;                                          // passes in a row that took no stream
        for (uint32_t pass = 0, headBefore = qHead; qHead < nRest; ++pass) {
            if (idle > 100000u) { snprintf(dev->err, sizeof dev->err, "coder launch: no arena slot in %u passes (queue at %u of %u)", idle, qHead, nRest); return -1; }
            if (idle) { struct timespec ts = {0, idle < 64u ? 200000L : 2000000L}; nanosleep(&ts, nullptr); }      // (a pass that took streams is followed at once)
            headBefore = qHead;
            ++tailLaunches;
            HIP_TRY(launch());
            HIP_TRY(hipEventRecord((hipEvent_t)dev->ev[1], st));
            HIP_TRY(hipMemcpyAsync(&qHead, dev->queueHead, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(sizes.data(), dev->dSizes, 4ull * nItems, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(restarts.data(), dev->dRestarts, 64ull * nItems, hipMemcpyDeviceToHost, st));
            HIP_TRY(wait_stream(dev, st));
            idle = qHead > headBefore ? 0u : idle + 1u;
        }
    }
    if (timing) {
        float a = 0;
        (void)hipEventElapsedTime(&a, (hipEvent_t)dev->ev[0], (hipEvent_t)dev->ev[1]);
        timing->encode_ms += a; timing->launches += 1 + tailLaunches; timing->items += nItems; timing->h2d_bytes += inputBytes;      // (every pass is a kernel of its own in a profiler's table: the launches counted here are what its average is over)
        if (ids) timing->id_strings += ids->n_strings;
        if (gather && gather->n_strings) {
            float g = 0; (void)hipEventElapsedTime(&g, (hipEvent_t)dev->ev[4], (hipEvent_t)dev->ev[5]);
            timing->gather_ms += g; timing->gather_symbols += gather->symbols;
            timing->gather_bytes += gather->qvz ? 4u * gather->symbols + (gather->symbols * 3u + 3u) / 4u + gather->symbols / 4u
                                                : (gather->bits == 6u ? gather->symbols + (gather->symbols * 3u + 3u) / 4u : 2u * gather->symbols + (gather->symbols * gather->bits + 7u) / 8u);
        }
        timing->tail_launches += tailLaunches;
        for (uint32_t i = 0; i < nItems; ++i) {
            timing->restarts += restarts[16ull * i];
            if (items[i].kind == KIND_PPMD) timing->max_restarts = std::max<uint64_t>(timing->max_restarts, restarts[16ull * i]);
            for (int k = 1; k < 16; ++k) timing->win[k] += restarts[16ull * i + k];
            const uint32_t len = writtenLen.empty() ? items[i].in_len : writtenLen[i];      // (what was coded, not the room a device-written stream was given)
            if (items[i].kind == KIND_PPMD) timing->ppmd_symbols += len; else timing->rc_symbols += len;
        }
    }
    return 0;
}

// Give every lane the device buffers of the best-equipped one.  Which slice a lane gets changes from batch to batch; a
// lane that had to grow a buffer in the middle of a batch would hipFree -- which waits for every kernel in flight, a
// second or more while the long streams are being coded.  Called between batches (nothing in flight).
int lanes_equalize(Device* const* lanes, size_t n)
{
    if (n < 2) return 0;
    size_t mIn = 0, mScratch = 0, mItems = 0, mOrder = 0, mSizes = 0, mRestarts = 0, mPlans = 0, mBlocks = 0;
    for (size_t i = 0; i < n; ++i) {
        const Device* d = lanes[i];
        mIn = std::max(mIn, d->capIn); mScratch = std::max(mScratch, d->capScratch); mItems = std::max(mItems, d->capItems); mOrder = std::max(mOrder, d->capOrder);
        mSizes = std::max(mSizes, d->capSizes); mRestarts = std::max(mRestarts, d->capRestarts); mPlans = std::max(mPlans, d->capPlans); mBlocks = std::max(mBlocks, d->capBlocks);
    }
    for (size_t i = 0; i < n; ++i) {
        Device* dev = lanes[i];
        HIP_TRY(hipSetDevice(dev->deviceId));
        for (uint32_t k = 0; k < dev->nOldStage; ++k) pinned_free(dev->oldStage[k], dev->oldStageCap[k], !dev->stagePageable);      // nothing is in flight here
        dev->nOldStage = 0;
        for (uint32_t k = 0; k < dev->nOldDev; ++k) (void)hipFree(dev->oldDev[k]);
        dev->nOldDev = 0;
        auto grow = [&](auto*& p, size_t& cap, size_t want) -> int {
            if (cap >= want && p) return 0;
            if (p) (void)hipFree(p);
            p = nullptr; cap = 0;
            HIP_TRY(hipMalloc((void**)&p, want));
            cap = want; return 0;
        };
        if (grow(dev->dIn, dev->capIn, mIn) || grow(dev->dScratch, dev->capScratch, mScratch) || grow(dev->dItems, dev->capItems, mItems) || grow(dev->dOrder, dev->capOrder, mOrder) ||
            grow(dev->dSizes, dev->capSizes, mSizes) || grow(dev->dRestarts, dev->capRestarts, mRestarts) || grow(dev->dPlans, dev->capPlans, mPlans) || grow(dev->dBlocks, dev->capBlocks, mBlocks)) return -1;
    }
    return 0;
}

int gather_quality_raw(Device* dev, const uint8_t* input, size_t inputBytes, const GatherPlan& plan, std::vector<uint8_t>& out, BatchTiming* timing)
{
    HIP_TRY(hipSetDevice(dev->deviceId));
    hipStream_t st = (hipStream_t)dev->stream;
    const uint64_t gatherBase = ((uint64_t)inputBytes + 15u) & ~15ull;
    if (plan.qvz) { snprintf(dev->err, sizeof dev->err, "the QVZ gather runs inside a batch (its tables travel with the streams)"); return -1; }
    const size_t descBytes = plan.bits == 6u ? sizeof(QuaString) : sizeof(QuaPairString);
    if (plan.desc_off & 15u || plan.desc_off + (uint64_t)plan.n_strings * descBytes > inputBytes || (plan.bits != 6u && plan.bits != 3u && plan.bits != 1u) ||
        (plan.bits != 6u && plan.n_list_off + plan.n_list_bytes > inputBytes)) { snprintf(dev->err, sizeof dev->err, "quality gather plan outside the input"); return -1; }
    if (plan.bits == 6u) {
        const QuaString* qs = (const QuaString*)(input + plan.desc_off);
        for (uint32_t i = 0; i < plan.n_strings; ++i)
            if ((qs[i].src_bit >> 3) + (6ull * qs[i].len + 7u) / 8u + 4u > plan.desc_off || (uint64_t)qs[i].dst_off + qs[i].len > plan.out_bytes) { snprintf(dev->err, sizeof dev->err, "quality string %u outside the input", i); return -1; }
    } else {
        const QuaPairString* qs = (const QuaPairString*)(input + plan.desc_off);
        for (uint32_t i = 0; i < plan.n_strings; ++i)
            if ((qs[i].src_bit >> 3) + ((uint64_t)plan.bits * qs[i].len + 7u) / 8u + 2u > plan.desc_off || qs[i].n_count > qs[i].len || 2ull * ((uint64_t)qs[i].dst_off + qs[i].len - qs[i].n_count) > plan.out_bytes ||
                (uint64_t)qs[i].n_off + qs[i].n_count > plan.n_list_bytes) { snprintf(dev->err, sizeof dev->err, "quality string %u outside the input", i); return -1; }
    }
    if (ensure(dev, dev->dIn, dev->capIn, gatherBase + plan.out_bytes + kInSlack)) return -1;
    HIP_TRY(hipMemcpyAsync(dev->dIn, input, inputBytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord((hipEvent_t)dev->ev[4], st));
    const uint32_t blocks = std::max(1u, std::min<uint32_t>((plan.n_strings + 3u) / 4u, (uint32_t)dev->cus * 16u));
    if (plan.bits == 6u)
        hipLaunchKernelGGL(fs_gather_quality, dim3(blocks), dim3(256), 0, st, (const QuaString*)(dev->dIn + plan.desc_off), plan.n_strings,
                           (const uint8_t*)dev->dIn, (uint8_t*)(dev->dIn + gatherBase));
    else
        hipLaunchKernelGGL(fs_gather_quality_pairs, dim3(blocks), dim3(256), 0, st, (const QuaPairString*)(dev->dIn + plan.desc_off), plan.n_strings,
                           (const uint8_t*)dev->dIn, (const uint8_t*)(dev->dIn + plan.n_list_off), (uint8_t*)(dev->dIn + gatherBase), plan.bits, plan.sym_of_bit[0], plan.sym_of_bit[1]);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord((hipEvent_t)dev->ev[5], st));
    out.resize(plan.out_bytes);
    HIP_TRY(hipMemcpyAsync(out.data(), dev->dIn + gatherBase, plan.out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(wait_stream(dev, st));
    if (timing) { float g = 0; (void)hipEventElapsedTime(&g, (hipEvent_t)dev->ev[4], (hipEvent_t)dev->ev[5]); timing->gather_ms += g; timing->gather_symbols += plan.symbols; timing->gather_bytes += plan.symbols + (plan.symbols * 3u + 3u) / 4u; }
    return 0;
}

// fs_tokenise_ids on its own (parity checks): `input` holds the packed headers, field tables, jobs and strings of `plan`; the
// two streams of job j come back in tok[j] / val[j]
int tokenise_ids_raw(Device* dev, const uint8_t* input, size_t inputBytes, const IdPlan& plan, std::vector<std::vector<uint8_t>>& tok, std::vector<std::vector<uint8_t>>& val)
{
    HIP_TRY(hipSetDevice(dev->deviceId));
    hipStream_t st = (hipStream_t)dev->stream;
    const uint64_t gatherBase = ((uint64_t)inputBytes + 15u) & ~15ull;
    if (plan.jobs_off & 7u || plan.strings_off & 7u || plan.jobs_off + (uint64_t)plan.n_jobs * sizeof(IdJob) > inputBytes || plan.strings_off + (uint64_t)plan.n_strings * sizeof(IdString) > inputBytes) {
        snprintf(dev->err, sizeof dev->err, "read-id plan outside the input"); return -1;
    }
    const IdJob* jb = (const IdJob*)(input + plan.jobs_off); const IdString* ss = (const IdString*)(input + plan.strings_off);
    std::vector<StreamItem> items(2 * (size_t)plan.n_jobs); memset(items.data(), 0, items.size() * sizeof(StreamItem));
    for (uint32_t j = 0; j < plan.n_jobs; ++j) {
        bool ok = (uint64_t)jb[j].first + jb[j].count <= plan.n_strings && jb[j].tok_item == 2 * j && jb[j].val_item == 2 * j + 1 && jb[j].table_off + 8u <= inputBytes && (jb[j].table_off & 7u) == 0 &&
                  jb[j].tok_out <= plan.out_bytes && jb[j].val_out <= plan.out_bytes;
        for (uint32_t k = 0; ok && k < jb[j].count; ++k) { const IdString& s = ss[jb[j].first + k]; ok = s.len <= 255u && id_string_inside(s.src_bit, s.len, inputBytes); }
        if (!ok) { snprintf(dev->err, sizeof dev->err, "read-id job %u outside the input", j); return -1; }
    }
    IdLaunch idl;
    id_launch_plan(jb, plan.n_jobs, plan.n_strings, gatherBase + plan.out_bytes, idl);
    if (ensure(dev, dev->dIn, dev->capIn, gatherBase + plan.out_bytes + idl.bytes + kInSlack)) return -1;
    if (ensure(dev, dev->dItems, dev->capItems, sizeof(StreamItem) * items.size() + 64)) return -1;
    HIP_TRY(hipMemcpyAsync(dev->dIn, input, inputBytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dev->dItems, items.data(), sizeof(StreamItem) * items.size(), hipMemcpyHostToDevice, st));
    if (plan.n_jobs && id_launch(dev, st, plan, gatherBase, idl)) return -1;
    std::vector<uint8_t> out(plan.out_bytes + 16);
    HIP_TRY(hipMemcpyAsync(out.data(), dev->dIn + gatherBase, plan.out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(items.data(), dev->dItems, sizeof(StreamItem) * items.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(wait_stream(dev, st));
    tok.assign(plan.n_jobs, {}); val.assign(plan.n_jobs, {});
    for (uint32_t j = 0; j < plan.n_jobs; ++j) {
        const uint64_t nt = 2ull * items[2 * j].in_len, nv = 2ull * items[2 * j + 1].in_len;
        if (jb[j].tok_out + nt > plan.out_bytes || jb[j].val_out + nv > plan.out_bytes) { snprintf(dev->err, sizeof dev->err, "read-id job %u wrote outside its streams", j); return -1; }
        tok[j].assign(out.begin() + jb[j].tok_out, out.begin() + jb[j].tok_out + nt); val[j].assign(out.begin() + jb[j].val_out, out.begin() + jb[j].val_out + nv);
    }
    return 0;
}

// Entropy-code independent streams and hand the raw coded bytes back (layout: items[i].out_off).
int encode_streams_raw(Device* dev, const uint8_t* input, size_t inputBytes, std::vector<StreamItem>& items,
                       std::vector<uint8_t>& raw, std::vector<uint32_t>& sizes, BatchTiming* timing)
{
    raw.clear(); sizes.clear();
    if (items.empty()) return 0;
    uint64_t scratch = 0;
    if (run_encode(dev, input, inputBytes, items, sizes, scratch, timing)) return -1;
    raw.resize(scratch);
    HIP_TRY(hipMemcpy(raw.data(), dev->dScratch, scratch, hipMemcpyDeviceToHost));
    if (timing) timing->d2h_bytes += scratch;
    return 0;
}

// Entropy-code a batch of bins and assemble their blocks: `input` holds every stream's pre-entropy
// bytes / (symbol, ctx) pairs.
static uint64_t emit_stream_bytes(uint32_t c, uint32_t inLen)
{ return (c == ECH_COUNT || fsemit::is_bit_channel(c) || c == ECH_HARD || c == ECH_HARD_PE) ? (uint64_t)inLen : 2ull * inLen; }

int emit_streams_raw(Device* dev, const uint8_t* input, size_t inputBytes, const EmitPlan& plan, std::vector<StreamItem>& items, std::vector<std::vector<std::vector<uint8_t>>>& streams)
{
    HIP_TRY(hipSetDevice(dev->deviceId));
    hipStream_t st = (hipStream_t)dev->stream;
    if (emit_check(dev, input, inputBytes, plan, (uint32_t)items.size())) return -1;
    const uint64_t emitBase = ((uint64_t)inputBytes + 15u) & ~15ull;
    EmitLaunch eml; emit_launch_plan(plan, emitBase + plan.out_bytes, eml);
    if (ensure(dev, dev->dIn, dev->capIn, emitBase + plan.out_bytes + eml.bytes + kInSlack)) return -1;
    if (ensure(dev, dev->dItems, dev->capItems, sizeof(StreamItem) * items.size() + 64)) return -1;
    HIP_TRY(hipMemcpyAsync(dev->dIn, input, inputBytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dev->dItems, items.data(), sizeof(StreamItem) * items.size(), hipMemcpyHostToDevice, st));
    if (emit_launch(dev, st, plan, emitBase, eml)) return -1;
    std::vector<uint8_t> out(plan.out_bytes + 16);
    HIP_TRY(hipMemcpyAsync(out.data(), dev->dIn + emitBase, plan.out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(items.data(), dev->dItems, sizeof(StreamItem) * items.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(wait_stream(dev, st));
    const EmitJob* jobs = (const EmitJob*)(input + plan.jobs_off);
    streams.assign(plan.n_jobs, std::vector<std::vector<uint8_t>>(ECH_COUNT + 1));
    for (uint32_t j = 0; j < plan.n_jobs; ++j)
        for (uint32_t c = 0; c <= ECH_COUNT; ++c) {
            if (jobs[j].item[c] == 0xFFFFFFFFu) continue;
            const uint64_t n = emit_stream_bytes(c, items[jobs[j].item[c]].in_len);
            if (jobs[j].out_off[c] + n > plan.out_bytes) { snprintf(dev->err, sizeof dev->err, "emission job %u: stream %u longer than its room", j, c); return -1; }
            streams[j][c].assign(out.begin() + (ptrdiff_t)jobs[j].out_off[c], out.begin() + (ptrdiff_t)(jobs[j].out_off[c] + n));
        }
    return 0;
}

int encode_batch(Device* dev, const uint8_t* input, size_t inputBytes, std::vector<StreamItem>& items,
                 std::vector<BlockPlan>& plans, std::vector<uint8_t>& blocks, std::vector<uint64_t>& blockSizes,
                 BatchTiming* timing, const GatherPlan* gather, const IdPlan* ids, const EmitPlan* emit)
{
    const uint32_t nItems = (uint32_t)items.size(), nBins = (uint32_t)plans.size();
    blockSizes.assign(nBins, 0);                 // `blocks` keeps its size between calls: resize() below does not re-zero what is overwritten anyway
    if (nItems == 0) { blocks.clear(); return 0; }
    std::vector<uint32_t> sizes; uint64_t scratch = 0;
    if (run_encode(dev, input, inputBytes, items, sizes, scratch, timing, gather, ids, emit)) return -1;
    hipStream_t st = (hipStream_t)dev->stream;
    if (ensure(dev, dev->dPlans, dev->capPlans, sizeof(BlockPlan) * nBins)) return -1;
    // a stream that filled its scratch slot was clipped: refuse rather than emit a corrupt block
    for (uint32_t i = 0; i < nItems; ++i)
        if (sizes[i] == 0xFFFFFFFFu) {
            snprintf(dev->err, sizeof dev->err, "stream item %u: symbol or context outside its coder's alphabet (corrupted input)", i);
            return -2;
        } else if (sizes[i] >= items[i].out_cap && items[i].in_len > 0) {
            snprintf(dev->err, sizeof dev->err, "stream item %u overflowed its output slot (%u >= %u)", i, sizes[i], items[i].out_cap);
            return -2;
        }
    uint64_t total = 0;
    for (uint32_t b = 0; b < nBins; ++b) {
        BlockPlan& pl = plans[b];
        uint64_t sz = 42ull + 16ull * pl.n_streams + 1ull;
        for (uint32_t s = 0; s < pl.n_streams; ++s) sz += sizes[pl.first_item + s];
        pl.block_off = total; blockSizes[b] = sz; total += sz;
    }
    if (ensure(dev, dev->dBlocks, dev->capBlocks, total + 16)) return -1;
    HIP_TRY(hipMemcpyAsync(dev->dPlans, plans.data(), sizeof(BlockPlan) * nBins, hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord((hipEvent_t)dev->ev[2], st));
    hipLaunchKernelGGL(fs_assemble_blocks, dim3(nBins), dim3(256), 0, st, (const BlockPlan*)dev->dPlans, (const StreamItem*)dev->dItems,
                       (const uint32_t*)dev->dSizes, (const uint8_t*)dev->dScratch, (uint8_t*)dev->dBlocks);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord((hipEvent_t)dev->ev[3], st));
    blocks.resize(total);
    HIP_TRY(hipMemcpyAsync(blocks.data(), dev->dBlocks, total, hipMemcpyDeviceToHost, st));
    HIP_TRY(wait_stream(dev, st));
    if (timing) {
        float b = 0;
        (void)hipEventElapsedTime(&b, (hipEvent_t)dev->ev[2], (hipEvent_t)dev->ev[3]);
        timing->assemble_ms += b; timing->d2h_bytes += total;
    }
    return 0;
}

}  // namespace fsengine
