// Device-side read matcher: the LZ-window search of ReadsClassifierSE::ConstructMatchTree
// (/root/reference/fastore/fastore_pack/ReadsClassifier.cpp:95-442: PrepareLzBuffer :38-52, FindBestLzMatch :55-83,
// UpdateLzMatchResult ReadsClassifier.h:160-196) for every read of every match-tree construction of a bin -- the
// top-level one over the sorted reads and one per stored sub-tree with its copied root.
//
// What the reference does per read, in processing order: drop the oldest of the W window slots, compare the read with
// every slot front to back (cost = |shift| * shiftCost + mismatches * mismatchCost over the overlap, shift = difference of
// the signature positions, |shift| <= 127), keep the cheapest slot at or below the threshold -- the first one among equal
// costs --, then put the read at the front, or, if it is an exact duplicate of a slot that is not the sub-tree's root
// copy, at the back, from where the next read drops it unseen.  So the window a read sees is the last W-1 non-duplicate
// reads before it (root copy included), newest first, padded with dummy slots of 256 x 'N' while fewer exist.
//
// MI355X mapping: one workgroup per construction, ONE THREAD PER WINDOW SLOT.  A thread keeps its slot's read in
// registers as three bit planes (two base bits and an 'N' flag per position); the current read's planes are wave-uniform
// (scalar loads).  Every thread prices its slot -- a per-lane 256-bit funnel shift by the difference of the signature
// positions, XOR, population count -- and the workgroup takes the minimum of (cost, slot age): a DPP min inside each
// wave, one LDS exchange and one barrier across the waves.  The thread that owns the oldest slot then takes the read
// over.  Duplicates, the root copy and the dummy slots follow the reference to the letter, so the table that comes back
// -- matched read, cost, shift, exact-duplicate flag per read -- is what the host's serial scan computes (checked read by
// read: fsgpu_matcher_check, tests/test_gpu.py).  The prefix-buffer search of -r/-l (ReadsClassifier.cpp:115-153,
// 329-392), which needs the final decisions of all earlier reads, stays with the host tree builder that consumes the table.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "device_types.h"
#include "engine.h"
#include "mates_core.h"

using namespace fsdev;

namespace {

enum : uint32_t { kNone = 0xFFFFFFFFu };

// ASCII bases -> four bit planes of FW words each in a frame ALIGNED AT THE SIGNATURE: bit OFF + p - minPos stands for
// base p, so two reads of a bin line up bit for bit whatever their shift (OFF = 32 * (FW / 2): 160 for reads of up to 160
// bases, 256 up to 256).  Planes: the two bits of (c >> 1) & 3 (A 0, C 1, T 2, G 3; 0 for 'N'), the 'N' flag, and
// "a base is here".  Two bases differ where both are present and any of the first three planes differs.
template <int FW> __global__ __launch_bounds__(256) void fs_pack_bases(const uint8_t* __restrict__ seq, const MatchRead* __restrict__ reads, uint32_t nReads,
                                                                        uint32_t* __restrict__ planes)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const uint32_t r = g / FW, w = g % FW;
    if (r >= nReads) return;
    const MatchRead rd = reads[r];
    const uint8_t* s = seq + rd.seq_off;
    const int32_t OFF = 32 * (FW / 2);
    uint32_t p0 = 0, p1 = 0, pn = 0, pv = 0;
    for (uint32_t b = 0; b < 32u; ++b) {
        const int32_t pos = (int32_t)(32u * w + b) - OFF + (int32_t)rd.min_pos;
        if (pos < 0 || pos >= (int32_t)rd.len) continue;
        const uint32_t c = s[pos];
        const bool acgt = c == 'A' || c == 'C' || c == 'G' || c == 'T';
        const uint32_t code = acgt ? (c >> 1) & 3u : 0u;
        p0 |= (code & 1u) << b; p1 |= (code >> 1) << b; pn |= (acgt ? 0u : 1u) << b; pv |= 1u << b;
    }
    uint32_t* o = planes + (size_t)r * (4 * FW);
    o[w] = p0; o[FW + w] = p1; o[2 * FW + w] = pn; o[3 * FW + w] = pv;
}

// The same planes straight from the bin file's packed bases (the reader of fastore_bin/FastqPacker.cpp:290-411,
// IFastqPacker::ReadNextRecord: a read's bases MSB first, two bits each -- index into the archive's symbol order -- or three
// when the read holds an 'N'; the signature's bases are not stored, they are the base-4 digits of the signature at sig_pos).
// One thread per plane word again: its 32 bases are at most 96 consecutive bits of the stream.  No ASCII in between: the code
// of a base goes through the symbol order to the same three plane bits fs_pack_bases takes from the character.
// order: the five symbols of the archive (MinimizerParameters::dnaSymbolOrder), a byte each in two words.
template <int FW> __global__ __launch_bounds__(256) void fs_unpack_planes(const uint8_t* __restrict__ dna, const PackedRead* __restrict__ packed, const MatchRead* __restrict__ reads,
                                                                           uint32_t nReads, uint32_t order0, uint32_t order1, uint32_t sigLen, uint32_t* __restrict__ planes)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const uint32_t r = g / FW, w = g % FW;
    if (r >= nReads) return;
    const MatchRead rd = reads[r];
    const PackedRead pk = packed[r];
    const bool plain = (pk.info & PACKED_PLAIN) != 0u, hasSig = (pk.info & PACKED_HAS_SIG) != 0u;
    const uint32_t sigId = pk.info & ((1u << PACKED_SIG_BITS) - 1u), sigPos = (pk.info >> PACKED_SIG_BITS) & 0xFFu, hole = hasSig ? sigLen : 0u;
    const uint32_t bits = plain ? 2u : 3u;
    const int32_t OFF = 32 * (FW / 2);
    uint32_t p0 = 0, p1 = 0, pn = 0, pv = 0;
    for (uint32_t b = 0; b < 32u; ++b) {
        const int32_t pos = (int32_t)(32u * w + b) - OFF + (int32_t)rd.min_pos;
        if (pos < 0 || pos >= (int32_t)rd.len) continue;
        uint32_t code;
        if (hasSig && (uint32_t)pos >= sigPos && (uint32_t)pos < sigPos + hole) code = (sigId >> (2u * (hole - 1u - ((uint32_t)pos - sigPos)))) & 3u;
        else {
            const uint32_t j = (uint32_t)pos < sigPos || !hasSig ? (uint32_t)pos : (uint32_t)pos - hole;
            const uint64_t at = (uint64_t)pk.bit_off + (uint64_t)bits * j;
            const uint8_t* q = dna + (at >> 3);
            const uint32_t two = ((uint32_t)q[0] << 8) | (uint32_t)q[1];                  // (the buffer ends with spare bytes)
            code = (two >> (16u - (uint32_t)(at & 7u) - bits)) & ((1u << bits) - 1u);
        }
        const uint32_t c = code < 4u ? (order0 >> (8u * code)) & 0xFFu : (code == 4u ? order1 & 0xFFu : 0u);
        const bool acgt = c == 'A' || c == 'C' || c == 'G' || c == 'T';
        const uint32_t pc = acgt ? (c >> 1) & 3u : 0u;
        p0 |= (pc & 1u) << b; p1 |= (pc >> 1) << b; pn |= (acgt ? 0u : 1u) << b; pv |= 1u << b;
    }
    uint32_t* o = planes + (size_t)r * (4 * FW);
    o[w] = p0; o[FW + w] = p1; o[2 * FW + w] = pn; o[3 * FW + w] = pv;
}

// FS_UNPACK_CHECK=1: both ways of making the planes, word by word
__global__ void fs_compare_words(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, size_t n, uint32_t* __restrict__ differing)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicAdd(differing, 1u);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x)
{
    // DPP min scan: row_shr 1/2/4/8 inside each row of 16, row_bcast 15 / 31 across the rows; the wave's minimum ends in lane 63
    int v = (int)x;
    #define FS_MIN_STEP(ctrl, rowmask) do { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp((int)kNone, v, ctrl, rowmask, 0xf, false); \
                                            v = (int)(((uint32_t)v < o_) ? (uint32_t)v : o_); } while (0)
    FS_MIN_STEP(0x111, 0xf); FS_MIN_STEP(0x112, 0xf); FS_MIN_STEP(0x114, 0xf); FS_MIN_STEP(0x118, 0xf);
    FS_MIN_STEP(0x142, 0xa); FS_MIN_STEP(0x143, 0xc);
    #undef FS_MIN_STEP
    return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}

struct Shared {
    uint32_t key[2][16], match[2][16], info[2][16];     // per wave: its best key, the matched read, len | shift << 16 | noMismatches << 31; two buffers by parity
};

// MULTI: more than one wavefront per construction (windows of more than 64 slots)
template <int FW, bool MULTI> __global__ __launch_bounds__(MULTI ? 1024 : 64) void fs_match_reads(
    const MatchCall* __restrict__ calls, const uint32_t* __restrict__ callIds, const MatchRead* __restrict__ reads, const uint32_t* __restrict__ planes,
    const uint32_t* __restrict__ warm, MatchParams par, MatchRow* __restrict__ rows)
{
    __shared__ Shared sh;
    const MatchCall call = calls[callIds[blockIdx.x]];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, nWaves = blockDim.x >> 6;
    const uint32_t cap = par.window - 1u;                 // real slots: the reference drops one slot before every search
    constexpr int C0 = FW / 2 - 2, C1 = FW / 2 + 2;       // the four words around the signature (64 bases either side of its start): compared first
    // this thread's slot
    uint32_t e0[FW], e1[FW], en[FW], ev[FW];
    uint32_t eRead = kNone, eLen = 0, eMin = 0, ePush = 0; bool valid = false;
    uint32_t pushed = 0, nextSlot = 0;                     // entries put at the front so far; entry k lives in slot k % cap = nextSlot
    auto load = [&](uint32_t r) {
        const uint32_t* p = planes + (size_t)r * (4 * FW);
        _Pragma("unroll") for (int i = 0; i < FW; ++i) { e0[i] = p[i]; e1[i] = p[FW + i]; en[i] = p[2 * FW + i]; ev[i] = p[3 * FW + i]; }
        eRead = r; eLen = reads[r].len; eMin = reads[r].min_pos; valid = true;
    };
    auto take = [&](uint32_t r) {                          // the owner of the next slot takes read r over
        if (tid == nextSlot) { load(r); ePush = pushed; }
        ++pushed; nextSlot = nextSlot + 1u == cap ? 0u : nextSlot + 1u;
    };
    if (call.aux >= 0) take((uint32_t)call.aux);
    if (call.warm_count) {                                 // a later piece of a long construction: the window as it stands, one slot per thread
        if (tid < call.warm_count) { load(warm[call.warm_first + tid]); ePush = tid; }
        pushed = call.warm_count; nextSlot = call.warm_count == cap ? 0u : call.warm_count;
    }
    for (uint32_t i = 0; i < call.count; ++i) {
        const uint32_t r = call.first + i;
        const MatchRead rd = reads[r];
        const uint32_t rLen = rd.len, rMin = rd.min_pos;
        const int32_t thr = par.encode_threshold ? par.encode_threshold : (int32_t)(rLen / 2u);
        const uint32_t* rp = planes + (size_t)r * (4 * FW);          // wave-uniform address: scalar loads
        uint32_t key = kNone, myInfo = 0;
        {
            const int32_t shift = (int32_t)eMin - (int32_t)rMin;
            const uint32_t ashift = (uint32_t)(shift < 0 ? -shift : shift);
            const int32_t insertCost = (int32_t)ashift * par.shift_cost;
            bool live = valid && ashift <= 127u && insertCost <= thr;
            // stage 1: the words around the signature; a wave none of whose slots can still get to the threshold stops here
            uint32_t mism = 0;
            _Pragma("unroll") for (int k = C0; k < C1; ++k) {
                const uint32_t d = ((e0[k] ^ rp[k]) | (e1[k] ^ rp[FW + k]) | (en[k] ^ rp[2 * FW + k])) & ev[k] & rp[3 * FW + k];
                mism += (uint32_t)__builtin_popcount(d);
            }
            live = live && insertCost + (int32_t)mism * par.mismatch_cost <= thr;
            if (__ballot(live) != 0ull) {
                _Pragma("unroll") for (int k = 0; k < FW; ++k) {
                    if (k >= C0 && k < C1) continue;
                    const uint32_t d = ((e0[k] ^ rp[k]) | (e1[k] ^ rp[FW + k]) | (en[k] ^ rp[2 * FW + k])) & ev[k] & rp[3 * FW + k];
                    mism += (uint32_t)__builtin_popcount(d);
                }
                const int32_t cc = insertCost + (int32_t)mism * par.mismatch_cost;
                if (live && cc <= thr) {
                    key = ((uint32_t)cc << 16) | (pushed - 1u - ePush);       // age: 0 = newest
                    myInfo = eLen | (((uint32_t)shift & 0x7FFFu) << 16) | (mism == 0u ? 0x80000000u : 0u);
                }
            }
        }
        uint32_t myMatch = eRead;
        // the dummy slots (256 x 'N', signature position 0) are all alike and come behind the real ones: one thread prices them
        if (pushed < cap && tid == nextSlot && rMin <= 127u) {
            uint32_t real = 0;                                                // bases of the read from its signature position on that are not 'N'
            _Pragma("unroll") for (int k = FW / 2; k < FW; ++k) real += (uint32_t)__builtin_popcount(rp[3 * FW + k] & ~rp[2 * FW + k]);
            const uint32_t minLen = rLen - rMin < 256u ? rLen - rMin : 256u;
            const int32_t cc = (int32_t)rMin * par.shift_cost + (int32_t)real * par.mismatch_cost;
            if (cc <= thr) {
                key = ((uint32_t)cc << 16) | pushed;
                myInfo = 256u | (((uint32_t)(-(int32_t)rMin) & 0x7FFFu) << 16) | (real == 0u ? 0x80000000u : 0u);
                myMatch = kNone - 1u;                                        // dummy
                (void)minLen;
            }
        }
        // minimum over the workgroup; the owner of the minimum publishes its slot
        uint32_t best = wave_min_u32(key), bMatch, bInfo;
        if (MULTI) {
            const uint32_t par2 = i & 1u;
            if (key == best && (best != kNone ? true : lane == 0u)) { sh.key[par2][wave] = best; sh.match[par2][wave] = myMatch; sh.info[par2][wave] = myInfo; }
            __syncthreads();
            uint32_t k2 = lane < nWaves ? sh.key[par2][lane] : kNone;
            best = wave_min_u32(k2);
            const uint64_t who = __ballot(lane < nWaves && k2 == best);
            const uint32_t w = (uint32_t)__builtin_ctzll(who | (1ull << 63));
            bMatch = sh.match[par2][w < nWaves ? w : 0u]; bInfo = sh.info[par2][w < nWaves ? w : 0u];
        } else {
            const uint64_t who = __ballot(key == best);
            const uint32_t l = (uint32_t)__builtin_ctzll(who);
            bMatch = (uint32_t)__builtin_amdgcn_readlane((int)myMatch, (int)l); bInfo = (uint32_t)__builtin_amdgcn_readlane((int)myInfo, (int)l);
        }
        MatchRow row; row.match = -1; row.cost = (int16_t)(thr + 1); row.shift = 0; row.no_mismatches = 0; row.identical = 0; row.dummy = 0; row.pad = 0;
        bool identical = false;
        if (best != kNone) {
            const bool dummy = bMatch == kNone - 1u;
            const uint32_t cost = best >> 16, bLen = bInfo & 0xFFFFu;
            int32_t sh15 = (int32_t)((bInfo >> 16) & 0x7FFFu); if (sh15 & 0x4000) sh15 -= 0x8000;
            row.match = dummy ? -2 : (int32_t)bMatch; row.cost = (int16_t)cost; row.shift = (int16_t)sh15;
            row.no_mismatches = (uint8_t)(bInfo >> 31); row.dummy = dummy ? 1 : 0;
            // an exact duplicate of a real slot that is not the sub-tree's root copy goes to the back of the window and is
            // dropped by the next read (ReadsClassifier.cpp:184-186, 305)
            identical = cost == 0u && bLen == rLen && !dummy && (int32_t)bMatch != call.aux;
            row.identical = identical ? 1 : 0;
        }
        if (tid == 0u) rows[r] = row;
        if (!identical) take(r);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// fs_match_mates -- the mate search of a paired-end bin: LzCompressorPE::CompressPair's history search
// (fastore_pack/FastqCompressor.cpp:4610-4959) with the minimizer sets of FastqCategorizerBase::FindMinimizers
// (fastore_bin/FastqCategorizer.cpp:109-151).
//
// What the reference does per pair, in the order the tree walk emits the records: the oldest of the W history entries leaves;
// the mate's valid signatures are collected from its two halves (set 1: positions [0, ~len/2), set 2: [len/2, ..) minus what
// set 1 holds; per signature its first position); every history entry that lists one of them (an entry lists up to four:
// the smallest of its own sets) is aligned at each of its FOUR stored positions against that signature's position (shift =
// entry position - mate position, |shift| <= 127, cost = |shift| * s + mismatches over the overlap * m); the cheapest
// alignment wins, the first one in the order (signature ascending, entries oldest first, stored position 0..3) among equals.
// At or below the threshold the mate is coded against that entry.  The mate then enters the history at the front -- or, if
// it matched with cost 0 and no mismatch, at the back, from where the next pair drops it unseen.
//
// MI355X mapping: one 1024-thread workgroup per bin, the pairs one after the other; mates_core.h says what a pair costs it and what was
// measured on the way there.  Rows come back as the host's serial search computes them (checked pair by pair: fsgpu_pe_matcher_check).
enum : uint32_t { kMateWindowMax = fsmate::kWindowMax };

__global__ __launch_bounds__(1024) void fs_match_mates(const MateJob* __restrict__ jobs, const MatePair* __restrict__ pairs, const uint8_t* __restrict__ seq,
                                                                   const uint32_t* __restrict__ validBits, MateParams par, MateRow* __restrict__ rows, uint32_t* __restrict__ hist)
{
#if defined(__HIP_DEVICE_COMPILE__)      // (the compiler's host pass only needs the kernel's name)
    __shared__ fsmate::Shared sh;
    fsmate::search_bin(sh, jobs[blockIdx.x], pairs, seq, validBits, par, rows, hist + (size_t)blockIdx.x * fsmate::hist_words(par.window));
#endif
}

// (a lane's search buffers may be carved out of ONE allocation: match_lane_reserve; a buffer that outgrows its part gets an allocation of its own,
// and only what is not inside the block is ever freed by itself)
static thread_local const uint8_t* t_blockLo = nullptr; static thread_local const uint8_t* t_blockHi = nullptr;
static bool insideBlock(const void* p) { return p && (const uint8_t*)p >= t_blockLo && (const uint8_t*)p < t_blockHi; }
template <class T> int ensureBuf(fsengine::Device* dev, T*& p, size_t& cap, size_t need)
{
    if (need <= cap && p) return 0;
    if (p && !insideBlock(p)) (void)hipFree(p);
    p = nullptr; cap = 0;
    const size_t want = need + need / 4 + 4096;
    hipError_t e = hipMalloc((void**)&p, want);
    if (e != hipSuccess) { snprintf(dev->err, sizeof dev->err, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return -1; }
    cap = want;
    return 0;
}

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(dev->err, sizeof dev->err, "%s failed: %s", #x, hipGetErrorString(e_)); return -1; } } while (0)

}  // namespace

namespace fsengine {

// The searches of all host threads go through TWO streams per GPU (each thread has its own buffers; a thread's copies and
// kernels keep their order on the stream they share with others).  The coder lanes hold 14 of the 16 hardware queues for
// kernels that run for a second: a stream that had to share one of THOSE queues would wait behind such a kernel.
enum { kMatchStreamsMax = 6 };
struct StreamPool { hipStream_t s[kMatchStreamsMax] = {}; int users = 0; unsigned next = 0; };
// streams the matcher lanes of a device share: two (the coder lanes need a hardware queue each)
static int matchStreams() { return 2; }
static std::mutex g_poolMx; static StreamPool g_pool[16];

// parity check of the device-side unpack (fsgpu_unpack_check, or FS_UNPACK_CHECK=1 for a whole run): plane words compared / differing
std::atomic<bool> g_unpackCheck{getenv("FS_UNPACK_CHECK") && atoi(getenv("FS_UNPACK_CHECK")) != 0};
std::atomic<uint64_t> g_unpackChecked{0}, g_unpackDiffering{0};
void unpack_check(bool on) { g_unpackCheck = on; if (on) { g_unpackChecked = 0; g_unpackDiffering = 0; } }
void unpack_check_counts(uint64_t* words, uint64_t* differing) { *words = g_unpackChecked.load(); *differing = g_unpackDiffering.load(); }

// (polled between sleeps, not a blocking wait: engine.hip, wait_stream, says what a blocking wait has been seen to do)
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

static double laneClockMs() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6; }

struct MatchLane {
    int deviceId = 0;
    hipStream_t stream = nullptr; hipEvent_t evWait = nullptr, ev0 = nullptr, ev1 = nullptr;
    uint8_t* dSeq = nullptr; size_t capSeq = 0;
    MatchRead* dReads = nullptr; size_t capReads = 0;
    PackedRead* dPacked = nullptr; size_t capPacked = 0;      // (device-side unpack: dSeq then holds the bin's packed bases)
    MatchCall* dCalls = nullptr; size_t capCalls = 0;
    uint32_t* dIds = nullptr; size_t capIds = 0;
    uint32_t* dWarm = nullptr; size_t capWarm = 0;
    uint32_t* dPlanes = nullptr; size_t capPlanes = 0;
    MatchRow* dRows = nullptr; size_t capRows = 0;
    MatePair* dPairs = nullptr; size_t capPairs = 0; MateRow* dMateRows = nullptr; size_t capMateRows = 0; uint32_t* dValid = nullptr; size_t capValid = 0;
    uint32_t* dHist = nullptr; size_t capHist = 0;             // the mate searches' histories (mates_core.h: hist_words per bin of a launch)
    uint8_t* hStage = nullptr; size_t capStage = 0; bool stagePageable = false;
    bool ownStream = false;
    uint8_t* block = nullptr; size_t blockBytes = 0;      // the eight search buffers as made by match_lane_reserve: one allocation
    uint32_t calls = 0;                                   // searches so far (FS_TRACE times a lane's first)
};
// (ensureBuf's view of the lane whose buffers it is asked to grow)
struct BlockScope { explicit BlockScope(const MatchLane* m) { t_blockLo = m->block; t_blockHi = m->block ? m->block + m->blockBytes : nullptr; } ~BlockScope() { t_blockLo = t_blockHi = nullptr; } };

int match_lane_create(Device* dev, MatchLane** out, bool ownStream)
{
    *out = nullptr;
    HIP_TRY(hipSetDevice(dev->deviceId));
    MatchLane* m = new MatchLane(); m->deviceId = dev->deviceId;
    // (streams of normal priority: a high-priority queue makes the scheduler save and restore the resident coder waves
    // around every search -- measured: the coder kernels ran twice as long)
    hipError_t e = hipSuccess;
    if (ownStream) {
        e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
        m->ownStream = e == hipSuccess;
    } else {
        std::lock_guard<std::mutex> g(g_poolMx);
        StreamPool& p = g_pool[dev->deviceId & 15];
        for (int i = 0; i < matchStreams() && e == hipSuccess; ++i) if (!p.s[i]) {
            e = hipStreamCreateWithFlags(&p.s[i], hipStreamNonBlocking);
        }
        if (e == hipSuccess) { m->stream = p.s[p.next++ % (unsigned)matchStreams()]; ++p.users; }
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&m->evWait, hipEventBlockingSync | hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&m->ev0);
    if (e == hipSuccess) e = hipEventCreate(&m->ev1);
    if (e != hipSuccess) { snprintf(dev->err, sizeof dev->err, "matcher lane: %s", hipGetErrorString(e)); delete m; return -1; }
    *out = m;
    return 0;
}

void match_lane_destroy(MatchLane* m)
{
    if (!m) return;
    (void)hipSetDevice(m->deviceId);
    if (m->stream && m->ownStream) { (void)hipStreamSynchronize(m->stream); (void)hipStreamDestroy(m->stream); m->stream = nullptr; }
    if (m->stream) {
        (void)hipStreamSynchronize(m->stream);
        std::lock_guard<std::mutex> g(g_poolMx);
        StreamPool& p = g_pool[m->deviceId & 15];
        if (--p.users == 0) { for (int i = 0; i < kMatchStreamsMax; ++i) if (p.s[i]) { (void)hipStreamDestroy(p.s[i]); p.s[i] = nullptr; } }
        m->stream = nullptr;
    }
    void* ptrs[] = {m->dSeq, m->dReads, m->dPacked, m->dCalls, m->dIds, m->dWarm, m->dPlanes, m->dRows, m->dPairs, m->dMateRows, m->dValid, m->dHist};
    { BlockScope bs(m); for (void* p : ptrs) if (p && !insideBlock(p)) (void)hipFree(p); }
    if (m->block) (void)hipFree(m->block);
    if (m->hStage) pinned_free(m->hStage, m->capStage, !m->stagePageable);
    if (m->evWait) (void)hipEventDestroy(m->evWait);
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    delete m;
}

// Room for bins of up to maxReads table entries / maxSeqBytes bases / maxCalls constructions / maxWarm warm-up entries, made
// BEFORE the coder kernels of a batch are in flight: growing a buffer later would hipFree, which waits for every running kernel
int match_lane_reserve(Device* dev, MatchLane* m, size_t maxReads, size_t maxSeqBytes, size_t maxCalls, size_t maxWarm)
{
    HIP_TRY(hipSetDevice(m->deviceId));
    const double tr0 = dev->trace ? laneClockMs() : 0;
    const size_t planeBytes = maxReads * (size_t)(4 * 16) * 4u;
    BlockScope scope(m);
    {   // The eight buffers of a search as ONE allocation (a lane that has none yet: a fresh context -- the CLI -- makes its 24 matcher lanes at
        // once, side by side, and the runtime serialises their allocations: eight each were part of the 0.3 s the first slice of a cold
        // process is ready later than a warm one's).  A lane that has buffers grows them one by one, as before.
        const size_t need[8] = {maxSeqBytes + 64, maxReads * sizeof(MatchRead), maxReads * sizeof(PackedRead), maxCalls * sizeof(MatchCall), maxCalls * 4u, maxWarm * 4u + 16, planeBytes, maxReads * sizeof(MatchRow)};
        const bool fresh = !m->block && !m->dSeq && !m->dReads && !m->dPacked && !m->dCalls && !m->dIds && !m->dWarm && !m->dPlanes && !m->dRows;
        if (fresh) {
            size_t off[9]; off[0] = 0;
            for (int i = 0; i < 8; ++i) off[i + 1] = off[i] + ((need[i] + need[i] / 8 + 4096 + 255) & ~(size_t)255);
            uint8_t* blk = nullptr;
            if (hipMalloc((void**)&blk, off[8]) == hipSuccess) {
                m->block = blk; m->blockBytes = off[8];
                m->dSeq = blk + off[0]; m->capSeq = off[1] - off[0];
                m->dReads = (MatchRead*)(blk + off[1]); m->capReads = off[2] - off[1];
                m->dPacked = (PackedRead*)(blk + off[2]); m->capPacked = off[3] - off[2];
                m->dCalls = (MatchCall*)(blk + off[3]); m->capCalls = off[4] - off[3];
                m->dIds = (uint32_t*)(blk + off[4]); m->capIds = off[5] - off[4];
                m->dWarm = (uint32_t*)(blk + off[5]); m->capWarm = off[6] - off[5];
                m->dPlanes = (uint32_t*)(blk + off[6]); m->capPlanes = off[7] - off[6];
                m->dRows = (MatchRow*)(blk + off[7]); m->capRows = off[8] - off[7];
            } else (void)hipGetLastError();       // (no room for the whole: one by one below)
        }
    }
    if (ensureBuf(dev, m->dSeq, m->capSeq, maxSeqBytes + 64) || ensureBuf(dev, m->dReads, m->capReads, maxReads * sizeof(MatchRead)) || ensureBuf(dev, m->dPacked, m->capPacked, maxReads * sizeof(PackedRead)) ||
        ensureBuf(dev, m->dCalls, m->capCalls, maxCalls * sizeof(MatchCall)) || ensureBuf(dev, m->dIds, m->capIds, maxCalls * 4u) || ensureBuf(dev, m->dWarm, m->capWarm, maxWarm * 4u + 16) ||
        ensureBuf(dev, m->dPlanes, m->capPlanes, planeBytes) || ensureBuf(dev, m->dRows, m->capRows, maxReads * sizeof(MatchRow))) return -1;
    const double tr1 = dev->trace ? laneClockMs() : 0;
    const size_t upBytes = ((maxSeqBytes + 64 + 15) & ~(size_t)15) + ((maxReads * sizeof(PackedRead) + 15) & ~(size_t)15) + ((maxReads * sizeof(MatchRead) + 15) & ~(size_t)15) + ((maxCalls * sizeof(MatchCall) + 15) & ~(size_t)15) + ((maxCalls * 4u + 15) & ~(size_t)15) + maxWarm * 4u + 64;
    if (upBytes > m->capStage) {
        if (m->hStage) pinned_free(m->hStage, m->capStage, !m->stagePageable);
        m->hStage = nullptr; m->capStage = 0;
        size_t want = upBytes + 65536;
        m->stagePageable = false;      /* (a few megabytes: pinned in a one-shot context too -- a copy from pageable memory is the runtime's, synchronous, one thread at a time: 10-85 ms a call with 24 callers) */ m->hStage = (uint8_t*)pinned_alloc(&want, !m->stagePageable);
        if (!m->hStage) { snprintf(dev->err, sizeof dev->err, "device matcher: pinned staging buffer of %zu bytes failed", want); return -1; }
        m->capStage = want;
    }
    if (dev->trace && m->calls == 0) fprintf(stderr, "[trace] matcher lane reserve: device buffers %.1f ms, staging buffer %.1f ms\n", tr1 - tr0, laneClockMs() - tr1);
    return 0;
}

// One bin's searches.  seq: the bin's bases (ASCII); reads: every read of every construction, each construction's reads in
// processing order (a sub-tree's root copy is one more read, named by its call); rows[i] answers reads[i] (root copies: unset).
int match_reads(Device* dev, MatchLane* m, const uint8_t* seq, size_t seqBytes, const PackedDna* packed, const MatchRead* reads, size_t nReads,
                const MatchCall* calls, size_t nCalls, const uint32_t* warm, size_t nWarm, const MatchParams& par, MatchRow* rows, double* kernelMs)
{
    if (nReads == 0 || nCalls == 0) return 0;
    const bool timed = dev->trace && m->calls < 2u; ++m->calls;
    const double tc0 = timed ? laneClockMs() : 0;
    if (par.window < 2u || par.window > 1025u) { snprintf(dev->err, sizeof dev->err, "device matcher: window of %u slots not supported (2..1025)", par.window); return -1; }
    HIP_TRY(hipSetDevice(m->deviceId));
    uint32_t maxLen = 0;
    if (packed) {      // every read's stored bits inside the packed bytes: nothing the kernel reads is taken on trust
        if (packed->sig_len * 2u > PACKED_SIG_BITS || !packed->dna || !packed->reads) { snprintf(dev->err, sizeof dev->err, "device matcher: packed bases with a signature of %u bases", packed->sig_len); return -1; }
        for (size_t i = 0; i < nReads; ++i) {
            const PackedRead& q = packed->reads[i];
            const uint32_t hole = (q.info & PACKED_HAS_SIG) ? packed->sig_len : 0u, sigPos = (q.info >> PACKED_SIG_BITS) & 0xFFu;
            if (hole > reads[i].len || (hole && sigPos + hole > reads[i].len)) { snprintf(dev->err, sizeof dev->err, "device matcher: read %zu: signature outside the read", i); return -1; }
            const uint64_t endBit = (uint64_t)q.bit_off + (uint64_t)((q.info & PACKED_PLAIN) ? 2u : 3u) * (reads[i].len - hole);
            if (endBit > 8ull * packed->bytes) { snprintf(dev->err, sizeof dev->err, "device matcher: read %zu outside the packed bases", i); return -1; }
        }
    }
    const size_t srcBytes = packed ? packed->bytes : seqBytes;         // what goes up in dSeq
    for (size_t i = 0; i < nReads; ++i) {
        if ((uint64_t)reads[i].seq_off + reads[i].len > seqBytes || reads[i].min_pos > reads[i].len) { snprintf(dev->err, sizeof dev->err, "device matcher: read %zu outside the bases", i); return -1; }
        maxLen = std::max<uint32_t>(maxLen, reads[i].len);
    }
    if (maxLen > 256u) { snprintf(dev->err, sizeof dev->err, "device matcher: reads longer than 256 bases"); return -1; }
    const int FW = maxLen <= 160u ? 10 : 16;
    // constructions by the number of window slots they can fill: one wavefront, or one thread per slot up to the window
    std::vector<uint32_t> ids(nCalls), small, large[4];       // large: 128, 256, 512, 1024 threads
    const uint32_t cap = par.window - 1u;
    for (size_t i = 0; i < nWarm; ++i) if (warm[i] >= nReads) { snprintf(dev->err, sizeof dev->err, "device matcher: warm-up list outside the read table"); return -1; }
    for (size_t c = 0; c < nCalls; ++c) {
        const MatchCall& k = calls[c];
        if ((uint64_t)k.first + k.count > nReads || (k.aux >= 0 && (size_t)k.aux >= nReads)) { snprintf(dev->err, sizeof dev->err, "device matcher: construction %zu outside the read table", c); return -1; }
        if ((uint64_t)k.warm_first + k.warm_count > nWarm || k.warm_count > cap || (k.warm_count && k.aux >= 0)) { snprintf(dev->err, sizeof dev->err, "device matcher: warm-up list of construction %zu", c); return -1; }
        if (k.count == 0) continue;
        const uint32_t slots = std::min<uint32_t>(k.count + (k.aux >= 0 ? 1u : 0u) + k.warm_count, cap);
        if (slots <= 64u) small.push_back((uint32_t)c);
        else large[slots <= 128u ? 0 : (slots <= 256u ? 1 : (slots <= 512u ? 2 : 3))].push_back((uint32_t)c);
    }
    const size_t planeBytes = nReads * (size_t)(4 * FW) * 4u;
    BlockScope scope(m);
    if (ensureBuf(dev, m->dSeq, m->capSeq, srcBytes + 64) || ensureBuf(dev, m->dReads, m->capReads, nReads * sizeof(MatchRead)) || (packed && ensureBuf(dev, m->dPacked, m->capPacked, nReads * sizeof(PackedRead))) ||
        ensureBuf(dev, m->dCalls, m->capCalls, nCalls * sizeof(MatchCall)) || ensureBuf(dev, m->dIds, m->capIds, nCalls * 4u) || ensureBuf(dev, m->dWarm, m->capWarm, nWarm * 4u + 16) ||
        ensureBuf(dev, m->dPlanes, m->capPlanes, planeBytes) || ensureBuf(dev, m->dRows, m->capRows, nReads * sizeof(MatchRow))) return -1;
    // pinned staging for everything that goes up (the callers' arrays are pageable)
    const size_t upBytes = ((srcBytes + 64 + 15) & ~(size_t)15) + (packed ? (nReads * sizeof(PackedRead) + 15) & ~(size_t)15 : 0) + ((nReads * sizeof(MatchRead) + 15) & ~(size_t)15) + ((nCalls * sizeof(MatchCall) + 15) & ~(size_t)15) + ((nCalls * 4u + 15) & ~(size_t)15) + nWarm * 4u + 64;
    if (upBytes > m->capStage) {
        if (m->hStage) pinned_free(m->hStage, m->capStage, !m->stagePageable);
        m->hStage = nullptr; m->capStage = 0;
        size_t want = upBytes + upBytes / 4 + 65536;
        m->stagePageable = false;      /* (a few megabytes: pinned in a one-shot context too -- a copy from pageable memory is the runtime's, synchronous, one thread at a time: 10-85 ms a call with 24 callers) */ m->hStage = (uint8_t*)pinned_alloc(&want, !m->stagePageable);
        if (!m->hStage) { snprintf(dev->err, sizeof dev->err, "device matcher: pinned staging buffer of %zu bytes failed", want); return -1; }
        m->capStage = want;
    }
    const double tc1 = timed ? laneClockMs() : 0;
    uint8_t* h = m->hStage; size_t o = 0;
    memcpy(h + o, packed ? packed->dna : seq, srcBytes); memset(h + o + srcBytes, 0, 64); const size_t oSeq = o; o += (srcBytes + 64 + 15) & ~(size_t)15;      // (spare bytes: the unpack kernel reads two bytes per base)
    size_t oPacked = 0;
    if (packed) { memcpy(h + o, packed->reads, nReads * sizeof(PackedRead)); oPacked = o; o += (nReads * sizeof(PackedRead) + 15) & ~(size_t)15; }
    memcpy(h + o, reads, nReads * sizeof(MatchRead)); const size_t oReads = o; o += (nReads * sizeof(MatchRead) + 15) & ~(size_t)15;
    memcpy(h + o, calls, nCalls * sizeof(MatchCall)); const size_t oCalls = o; o += (nCalls * sizeof(MatchCall) + 15) & ~(size_t)15;
    uint32_t* hid = (uint32_t*)(h + o); const size_t oIds = o; size_t nid = 0;
    const size_t smallAt = nid; for (uint32_t c : small) hid[nid++] = c;
    size_t largeAt[4]; for (int g = 0; g < 4; ++g) { largeAt[g] = nid; for (uint32_t c : large[g]) hid[nid++] = c; }
    o += (nCalls * 4u + 15) & ~(size_t)15;
    const size_t oWarm = o;
    if (nWarm) memcpy(h + o, warm, nWarm * 4u);
    const double tc2 = timed ? laneClockMs() : 0;
    hipStream_t st = m->stream;
    HIP_TRY(hipMemcpyAsync(m->dSeq, h + oSeq, srcBytes + (packed ? 64 : 0), hipMemcpyHostToDevice, st));
    if (packed) HIP_TRY(hipMemcpyAsync(m->dPacked, h + oPacked, nReads * sizeof(PackedRead), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->dReads, h + oReads, nReads * sizeof(MatchRead), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->dCalls, h + oCalls, nCalls * sizeof(MatchCall), hipMemcpyHostToDevice, st));
    if (nid) HIP_TRY(hipMemcpyAsync(m->dIds, h + oIds, nid * 4u, hipMemcpyHostToDevice, st));
    if (nWarm) HIP_TRY(hipMemcpyAsync(m->dWarm, h + oWarm, nWarm * 4u, hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(m->ev0, st));
    const uint32_t packBlocks = (uint32_t)((nReads * (size_t)FW + 255) / 256);
    if (packed) {
        uint32_t o0 = 0, o1 = 0; memcpy(&o0, packed->symbol_order, 4); memcpy(&o1, packed->symbol_order + 4, 4);
        if (FW == 10) hipLaunchKernelGGL(fs_unpack_planes<10>, dim3(packBlocks), dim3(256), 0, st, (const uint8_t*)m->dSeq, (const PackedRead*)m->dPacked, (const MatchRead*)m->dReads, (uint32_t)nReads, o0, o1, packed->sig_len, m->dPlanes);
        else hipLaunchKernelGGL(fs_unpack_planes<16>, dim3(packBlocks), dim3(256), 0, st, (const uint8_t*)m->dSeq, (const PackedRead*)m->dPacked, (const MatchRead*)m->dReads, (uint32_t)nReads, o0, o1, packed->sig_len, m->dPlanes);
        HIP_TRY(hipGetLastError());
        if (g_unpackCheck.load()) {      // the planes once more from the ASCII bases (what the host unpacked), word by word against the device's own
            uint8_t* dAscii = nullptr; uint32_t* dPlanes2 = nullptr; uint32_t* dDiff = nullptr; uint32_t diff = 0;
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(hipMalloc((void**)&dAscii, seqBytes + 64)); HIP_TRY(hipMalloc((void**)&dPlanes2, planeBytes)); HIP_TRY(hipMalloc((void**)&dDiff, 4));
            HIP_TRY(hipMemcpy(dAscii, seq, seqBytes, hipMemcpyHostToDevice)); HIP_TRY(hipMemset(dDiff, 0, 4));
            if (FW == 10) hipLaunchKernelGGL(fs_pack_bases<10>, dim3(packBlocks), dim3(256), 0, st, (const uint8_t*)dAscii, (const MatchRead*)m->dReads, (uint32_t)nReads, dPlanes2);
            else hipLaunchKernelGGL(fs_pack_bases<16>, dim3(packBlocks), dim3(256), 0, st, (const uint8_t*)dAscii, (const MatchRead*)m->dReads, (uint32_t)nReads, dPlanes2);
            const size_t words = planeBytes / 4u;
            hipLaunchKernelGGL(fs_compare_words, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, st, (const uint32_t*)m->dPlanes, (const uint32_t*)dPlanes2, words, dDiff);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(hipMemcpy(&diff, dDiff, 4, hipMemcpyDeviceToHost));
            (void)hipFree(dAscii); (void)hipFree(dPlanes2); (void)hipFree(dDiff);
            g_unpackChecked += words; g_unpackDiffering += diff;
        }
    } else {
        if (FW == 10) hipLaunchKernelGGL(fs_pack_bases<10>, dim3(packBlocks), dim3(256), 0, st, (const uint8_t*)m->dSeq, (const MatchRead*)m->dReads, (uint32_t)nReads, m->dPlanes);
        else hipLaunchKernelGGL(fs_pack_bases<16>, dim3(packBlocks), dim3(256), 0, st, (const uint8_t*)m->dSeq, (const MatchRead*)m->dReads, (uint32_t)nReads, m->dPlanes);
        HIP_TRY(hipGetLastError());
    }
    auto launch = [&](size_t at, size_t n, uint32_t threads) {
        if (!n) return;
        const uint32_t* idp = m->dIds + at;
        if (threads == 64u) {
            if (FW == 10) hipLaunchKernelGGL((fs_match_reads<10, false>), dim3((uint32_t)n), dim3(64), 0, st, (const MatchCall*)m->dCalls, idp, (const MatchRead*)m->dReads, (const uint32_t*)m->dPlanes, (const uint32_t*)m->dWarm, par, m->dRows);
            else hipLaunchKernelGGL((fs_match_reads<16, false>), dim3((uint32_t)n), dim3(64), 0, st, (const MatchCall*)m->dCalls, idp, (const MatchRead*)m->dReads, (const uint32_t*)m->dPlanes, (const uint32_t*)m->dWarm, par, m->dRows);
        } else {
            if (FW == 10) hipLaunchKernelGGL((fs_match_reads<10, true>), dim3((uint32_t)n), dim3(threads), 0, st, (const MatchCall*)m->dCalls, idp, (const MatchRead*)m->dReads, (const uint32_t*)m->dPlanes, (const uint32_t*)m->dWarm, par, m->dRows);
            else hipLaunchKernelGGL((fs_match_reads<16, true>), dim3((uint32_t)n), dim3(threads), 0, st, (const MatchCall*)m->dCalls, idp, (const MatchRead*)m->dReads, (const uint32_t*)m->dPlanes, (const uint32_t*)m->dWarm, par, m->dRows);
        }
    };
    // the largest constructions first: they run the longest
    for (int g = 3; g >= 0; --g) launch(largeAt[g], large[g].size(), 128u << g);
    launch(smallAt, small.size(), 64u);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(m->ev1, st));
    const double tc3 = timed ? laneClockMs() : 0;
    HIP_TRY(hipMemcpyAsync(rows, m->dRows, nReads * sizeof(MatchRow), hipMemcpyDeviceToHost, st));
    const double tc4 = timed ? laneClockMs() : 0;
    HIP_TRY(hipEventRecord(m->evWait, st));
    HIP_TRY(wait_event_polled(m->evWait));
    if (timed) { float a = 0; (void)hipEventElapsedTime(&a, m->ev0, m->ev1);
                 fprintf(stderr, "[trace] matcher lane call %u (%zu reads): checks + buffers %.1f ms, staging copy %.1f ms, uploads + launches enqueued %.1f ms, rows' copy enqueued %.1f ms, waited %.1f ms (kernels %.1f ms)\n",
                         m->calls, nReads, tc1 - tc0, tc2 - tc1, tc3 - tc2, tc4 - tc3, laneClockMs() - tc4, a); }
    if (kernelMs) { float a = 0; (void)hipEventElapsedTime(&a, m->ev0, m->ev1); *kernelMs += a; }
    return 0;
}

// One paired-end bin's mate searches.  seq: the bin's bases (ASCII); pairs: its pairs in the order the tree walk emits them;
// validBits: one bit per signature (the archive's valid minimizers, 2^(2 sig_len) bits); rows[i] answers pairs[i].
int match_mates(Device* dev, MatchLane* m, const uint8_t* seq, size_t seqBytes, const MatePair* pairs, size_t nPairs, const uint32_t* validBits, size_t validWords,
                const MateParams& par, MateRow* rows, double* kernelMs)
{
    if (nPairs == 0) return 0;
    if (par.window < 1u || par.window > kMateWindowMax || par.sig_len < 2u || par.sig_len > 8u || validWords < ((1ull << (2u * par.sig_len)) + 31u) / 32u) {
        snprintf(dev->err, sizeof dev->err, "device mate search: window %u / signature length %u not supported", par.window, par.sig_len); return -1;
    }
    HIP_TRY(hipSetDevice(m->deviceId));
    for (size_t i = 0; i < nPairs; ++i)
        if ((uint64_t)pairs[i].mate_off + pairs[i].mate_len > seqBytes || pairs[i].mate_len > 255u || pairs[i].mate_len < par.sig_len) { snprintf(dev->err, sizeof dev->err, "device mate search: pair %zu outside the bases", i); return -1; }
    BlockScope scope(m);
    if (ensureBuf(dev, m->dSeq, m->capSeq, seqBytes + 64) || ensureBuf(dev, m->dPairs, m->capPairs, nPairs * sizeof(MatePair)) || ensureBuf(dev, m->dMateRows, m->capMateRows, nPairs * sizeof(MateRow)) ||
        ensureBuf(dev, m->dValid, m->capValid, 8192 + 64) || ensureBuf(dev, m->dCalls, m->capCalls, sizeof(MateJob) + 64) ||
        ensureBuf(dev, m->dHist, m->capHist, (size_t)fsmate::hist_words(par.window) * 4u)) return -1;
    hipStream_t st = m->stream;
    const MateJob job{0u, (uint32_t)nPairs};
    std::vector<uint32_t> vb(2048, 0u);
    memcpy(vb.data(), validBits, std::min<size_t>(validWords, 2048) * 4u);
    HIP_TRY(hipMemcpyAsync(m->dSeq, seq, seqBytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->dPairs, pairs, nPairs * sizeof(MatePair), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->dValid, vb.data(), 8192, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->dCalls, &job, sizeof job, hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(m->ev0, st));
    hipLaunchKernelGGL(fs_match_mates, dim3(1), dim3(fsmate::kThreads), 0, st, (const MateJob*)m->dCalls, (const MatePair*)m->dPairs, (const uint8_t*)m->dSeq, (const uint32_t*)m->dValid, par, m->dMateRows, m->dHist);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(m->ev1, st));
    HIP_TRY(hipMemcpyAsync(rows, m->dMateRows, nPairs * sizeof(MateRow), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(m->evWait, st));
    HIP_TRY(wait_event_polled(m->evWait));
    if (kernelMs) { float a = 0; (void)hipEventElapsedTime(&a, m->ev0, m->ev1); *kernelMs += a; }
    return 0;
}

// The mate searches of SEVERAL bins in one launch (a workgroup per bin; the bins of a batch share the archive's parameters): what lets
// the device search without the host threads waiting bin by bin.  jobs[j].rows[i] answers jobs[j].pairs[i].
int match_mates_batch(Device* dev, MatchLane* m, const MateBatchJob* jobs, size_t nJobs, const uint32_t* validBits, size_t validWords, const MateParams& par, double* kernelMs)
{
    if (nJobs == 0) return 0;
    if (par.window < 1u || par.window > kMateWindowMax || par.sig_len < 2u || par.sig_len > 8u || validWords < ((1ull << (2u * par.sig_len)) + 31u) / 32u) {
        snprintf(dev->err, sizeof dev->err, "device mate search: window %u / signature length %u not supported", par.window, par.sig_len); return -1;
    }
    HIP_TRY(hipSetDevice(m->deviceId));
    uint64_t seqTotal = 0, pairTotal = 0;
    std::vector<uint64_t> seqBase(nJobs);
    for (size_t j = 0; j < nJobs; ++j) {
        seqBase[j] = seqTotal; seqTotal += (jobs[j].seqBytes + 15u) & ~(uint64_t)15u; pairTotal += jobs[j].nPairs;
        for (size_t i = 0; i < jobs[j].nPairs; ++i) {
            const MatePair& p = jobs[j].pairs[i];
            if ((uint64_t)p.mate_off + p.mate_len > jobs[j].seqBytes || p.mate_len > 255u || p.mate_len < par.sig_len) { snprintf(dev->err, sizeof dev->err, "device mate search: job %zu, pair %zu outside the bases", j, i); return -1; }
        }
    }
    if (seqTotal > 0xFFFFFF00ull || pairTotal > 0xFFFFFF00ull) { snprintf(dev->err, sizeof dev->err, "device mate search: batch beyond 4 GiB"); return -1; }
    if (pairTotal == 0) return 0;
    BlockScope scope(m);
    if (ensureBuf(dev, m->dSeq, m->capSeq, seqTotal + 64) || ensureBuf(dev, m->dPairs, m->capPairs, pairTotal * sizeof(MatePair)) || ensureBuf(dev, m->dMateRows, m->capMateRows, pairTotal * sizeof(MateRow)) ||
        ensureBuf(dev, m->dValid, m->capValid, 8192 + 64) || ensureBuf(dev, m->dCalls, m->capCalls, nJobs * sizeof(MateJob) + 64) ||
        ensureBuf(dev, m->dHist, m->capHist, nJobs * (size_t)fsmate::hist_words(par.window) * 4u)) return -1;
    hipStream_t st = m->stream;
    std::vector<MateJob> mj(nJobs); std::vector<MatePair> all(pairTotal);
    uint64_t at = 0;
    for (size_t j = 0; j < nJobs; ++j) {
        mj[j] = MateJob{(uint32_t)at, (uint32_t)jobs[j].nPairs};
        for (size_t i = 0; i < jobs[j].nPairs; ++i) { all[at + i] = jobs[j].pairs[i]; all[at + i].mate_off += (uint32_t)seqBase[j]; }
        at += jobs[j].nPairs;
        if (jobs[j].seqBytes) HIP_TRY(hipMemcpyAsync(m->dSeq + seqBase[j], jobs[j].seq, jobs[j].seqBytes, hipMemcpyHostToDevice, st));
    }
    std::vector<uint32_t> vb(2048, 0u);
    memcpy(vb.data(), validBits, std::min<size_t>(validWords, 2048) * 4u);
    HIP_TRY(hipMemcpyAsync(m->dPairs, all.data(), pairTotal * sizeof(MatePair), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->dValid, vb.data(), 8192, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->dCalls, mj.data(), nJobs * sizeof(MateJob), hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(m->ev0, st));
    hipLaunchKernelGGL(fs_match_mates, dim3((uint32_t)nJobs), dim3(fsmate::kThreads), 0, st, (const MateJob*)m->dCalls, (const MatePair*)m->dPairs, (const uint8_t*)m->dSeq, (const uint32_t*)m->dValid, par, m->dMateRows, m->dHist);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(m->ev1, st));
    std::vector<MateRow> rows(pairTotal);
    HIP_TRY(hipMemcpyAsync(rows.data(), m->dMateRows, pairTotal * sizeof(MateRow), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(m->evWait, st));
    HIP_TRY(wait_event_polled(m->evWait));
    at = 0;
    for (size_t j = 0; j < nJobs; ++j) { if (jobs[j].nPairs) memcpy(jobs[j].rows, rows.data() + at, jobs[j].nPairs * sizeof(MateRow)); at += jobs[j].nPairs; }
    if (kernelMs) { float a = 0; (void)hipEventElapsedTime(&a, m->ev0, m->ev1); *kernelMs += a; }
    return 0;
}

}  // namespace fsengine
