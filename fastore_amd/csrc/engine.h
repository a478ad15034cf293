// Host-visible interface of the HIP batch engine (engine.hip).  No HIP types leak out.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>
#include "device_types.h"

namespace fsengine {

struct BatchTiming {
    double encode_ms = 0, assemble_ms = 0;      // HIP-event time of the two kernels (summed over launches)
    double gather_ms = 0; uint64_t gather_symbols = 0, gather_bytes = 0;   // fs_gather_quality: time, scores, bytes read + written
    uint64_t id_strings = 0;                                                 // fs_tokenise_ids: read ids tokenised on the device
    uint64_t launches = 0, items = 0, ppmd_symbols = 0, rc_symbols = 0, restarts = 0, max_restarts = 0;
    uint64_t h2d_bytes = 0, d2h_bytes = 0;
    uint64_t tail_launches = 0;                 // coder launches made again because no workgroup of the launch found a free arena slot (engine.hip: run_encode)
    uint64_t win[16] = {0};                     // windowed PPMd hit path, summed over the streams: [1] attempts [2] windows [3] symbols [4] rounds [5] redone; [8..15] phase clocks / 64
};

// What the lanes of one GPU share: the arena pool and its slot rings (see engine.hip).
struct Pool;

// One engine lane = one HIP stream with its own buffers.  Lanes of a GPU share the arena pool, so their kernels can
// be in flight together (the host front end of the next slice overlaps the device work of the previous ones).
struct Device {
    int deviceId, cus;
    char name[64];
    char err[256];
    void* stream;
    void* ev[6];
    void* evWait;                                          // blocking-sync event: a waiting lane thread sleeps instead of spinning
    Pool* pool; uint32_t* queueHead; uint32_t nWaves /* resident-wave cap */;
    uint8_t* dIn; size_t capIn;
    uint8_t* dScratch; size_t capScratch;
    void* dItems; size_t capItems;
    uint32_t* dOrder; size_t capOrder;
    uint32_t* dSizes; size_t capSizes;
    uint32_t* dRestarts; size_t capRestarts;
    void* dPlans; size_t capPlans;
    uint8_t* dBlocks; size_t capBlocks;
    uint8_t* hStage; size_t capStage;
    // staging buffers outgrown in the middle of a batch: un-registering pinned memory waits for every running kernel (as a
    // hipFree does), so they are kept until the lane is between batches (lanes_equalize) or goes
    void* oldStage[4]; size_t oldStageCap[4]; uint32_t nOldStage;
    // ... and device buffers outgrown in the middle of a batch (a lane that takes a second, larger slice): a hipFree waits for every running kernel
    // of the DEVICE -- of the other pipelines of a split pack too, whose streams run for seconds --, so they are kept as well
    void* oldDev[32]; uint32_t nOldDev;
    uint32_t stagePageable;                                // the staging buffers of this lane are not registered with the runtime
    uint32_t trace;                                        // FS_TRACE was set when the lane was made: every launch is waited for where it is made
};

int device_count();
int device_create(Device** out, int deviceId, uint32_t maxWaves, char* err, size_t errLen);
int lane_create(Device* first, Device** out, char* err, size_t errLen);      // another lane on the GPU (and pool) of `first`
void device_destroy(Device* dev);                                            // a lane; the pool goes with its last lane
// diagnostic: progress of the lane's current launch (work-queue head) and the pool's slot rings, read on a stream of its own
int lane_debug(Device* dev, char* out, size_t outLen);
// Pinned host memory whose pages are made by the CALLING thread (anonymous mapping, touched here) and then registered with
// the runtime: hipHostMalloc does both inside the runtime (~5 GB/s, and other threads' HIP calls queue behind it); the
// registration alone is 4-8 x shorter (tools/probes/hip_startup_probe.cpp).  `bytes` is rounded up; pinned_free wants the same value.
void* pinned_alloc(size_t* bytes, bool pin = true);      // pin = false: the pages only (pageable memory, the runtime stages the copies)
void pinned_free(void* p, size_t bytes, bool pinned = true);
// A context that packs once and goes (the CLI) keeps its lanes' staging buffers pageable: registering 2.6 GB of them and
// handing them back at exit costs more (~0.5 s of process time) than the runtime's own staged copies do (call before device_create).
void set_pageable_staging(bool on);
bool pageable_staging();
uint8_t* staging_buffer(Device* dev, size_t bytes);     // grow-only pinned host buffer for the batch input
void staging_release(Device* dev);                      // the lane's staging buffers go (a context that packs once: behind its last slice, beside the device's tail)
int lanes_equalize(Device* const* lanes, size_t n);     // between batches: every lane gets the device buffers of the best-equipped one
int encode_streams_raw(Device* dev, const uint8_t* input, size_t inputBytes, std::vector<fsdev::StreamItem>& items,
                       std::vector<uint8_t>& raw, std::vector<uint32_t>& sizes, BatchTiming* timing);
// Device-side read matcher (matcher.hip): a lane of its own (high-priority stream, own buffers) per caller; match_reads
// answers every read of one bin's match-tree constructions and returns when the rows are in `rows`
struct MatchLane;
int match_lane_create(Device* dev, MatchLane** out, bool ownStream = false);      // ownStream: a stream of its own instead of one of the searches' shared ones (the batched mate searches: their kernels run for a tenth of a second)
void match_lane_destroy(MatchLane* m);
int match_lane_reserve(Device* dev, MatchLane* m, size_t maxReads, size_t maxSeqBytes, size_t maxCalls, size_t maxWarm);
// packed != nullptr: the bases come as the bin file stores them and are unpacked on the device (fs_unpack_planes); `seq` then
// only travels for the FS_UNPACK_CHECK=1 comparison of the two ways
int match_reads(Device* dev, MatchLane* m, const uint8_t* seq, size_t seqBytes, const fsdev::PackedDna* packed, const fsdev::MatchRead* reads, size_t nReads,
                const fsdev::MatchCall* calls, size_t nCalls, const uint32_t* warm, size_t nWarm, const fsdev::MatchParams& par,
                fsdev::MatchRow* rows, double* kernelMs);
// parity check of fs_unpack_planes: while on, every packed search also builds the planes from `seq` and counts the words that differ
void unpack_check(bool on);
void unpack_check_counts(uint64_t* words, uint64_t* differing);
// the mate searches of one paired-end bin (matcher.hip: fs_match_mates)
int match_mates(Device* dev, MatchLane* m, const uint8_t* seq, size_t seqBytes, const fsdev::MatePair* pairs, size_t nPairs, const uint32_t* validBits, size_t validWords,
                const fsdev::MateParams& par, fsdev::MateRow* rows, double* kernelMs);
// ... of several bins in one launch (a workgroup per bin): jobs[j].rows[i] answers jobs[j].pairs[i]
struct MateBatchJob { const uint8_t* seq; size_t seqBytes; const fsdev::MatePair* pairs; size_t nPairs; fsdev::MateRow* rows; };
int match_mates_batch(Device* dev, MatchLane* m, const MateBatchJob* jobs, size_t nJobs, const uint32_t* validBits, size_t validWords, const fsdev::MateParams& par, double* kernelMs);
// fs_gather_quality on its own: `input` = packed scores then the descriptors (plan.desc_off); returns the gathered bytes
int gather_quality_raw(Device* dev, const uint8_t* input, size_t inputBytes, const fsdev::GatherPlan& plan, std::vector<uint8_t>& out, BatchTiming* timing);
// fs_tokenise_ids on its own (parity checks): tok[j] / val[j] = the (symbol, context) pair streams of job j (its items: 2 j, 2 j + 1)
int tokenise_ids_raw(Device* dev, const uint8_t* input, size_t inputBytes, const fsdev::IdPlan& plan, std::vector<std::vector<uint8_t>>& tok, std::vector<std::vector<uint8_t>>& val);
// the emission kernels on their own (parity checks): `input` holds the plan's jobs, ops, ids, bases and contig bytes; on return
// streams[j][c] = the bytes of channel c of job j as the kernels left them (c = fsdev::ECH_COUNT: the run-length coded LZ ids)
int emit_streams_raw(Device* dev, const uint8_t* input, size_t inputBytes, const fsdev::EmitPlan& plan, std::vector<fsdev::StreamItem>& items, std::vector<std::vector<std::vector<uint8_t>>>& streams);
// gather (optional): quality streams that fs_gather_quality writes behind the uploaded input (at inputBytes rounded up to
// 16) from the packed scores inside it; their items' in_off already point there
int encode_batch(Device* dev, const uint8_t* input, size_t inputBytes, std::vector<fsdev::StreamItem>& items,
                 std::vector<fsdev::BlockPlan>& plans, std::vector<uint8_t>& blocks, std::vector<uint64_t>& blockSizes,
                 BatchTiming* timing, const fsdev::GatherPlan* gather = nullptr, const fsdev::IdPlan* ids = nullptr, const fsdev::EmitPlan* emit = nullptr);

}  // namespace fsengine
