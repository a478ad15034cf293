// TEST-ONLY lock-step emulation of one 64-lane wavefront on the host.
//
// The coder cores (fastore_amd/csrc/*_core.h) are written for the device: one stream per wavefront, wave-uniform
// control flow, cross-lane steps through ballot / readlane / bpermute / DPP sums.  Compiled with -DFS_SIMT_EMU the
// same source runs here as 64 cooperative fibers, one per lane, that advance in a fixed cyclic order and meet at every
// cross-lane primitive -- so the 64-lane code paths (which the one-lane host build does not have) can be checked
// bit-for-bit against the oracle in a container without a GPU.  Never part of the product.
//
// Rules the emulation relies on (the device code follows them anyway): every lane reaches the same primitives in the
// same order (uniform control flow); a store whose lanes write different data is followed by FS_WAVE_SYNC() or by one
// of the primitives before another lane reads it.
#pragma once
#include <stdint.h>
#include <functional>

namespace simt {

enum { WAVE = 64 };
// run `body(lane)` on WAVE fibers in lock step; returns when all have returned
void run(const std::function<void(int)>& body);
// several waves of one workgroup side by side (they share memory, e.g. an LDS image): `body(wave, lane)`.  Each wave is
// in lock step with itself only; after every meeting point of a wave the other waves get their turn, so a wave that
// polls for another one's progress (through a primitive or barrier() per poll) cannot starve it.
void run_waves(int waves, const std::function<void(int, int)>& body);
int wave();
int lane();
void barrier();                                 // all lanes meet here
uint64_t ballot(bool p);
uint32_t readlane(uint32_t v, uint32_t srcLane);        // srcLane uniform
uint32_t bperm(uint32_t v, uint32_t srcLane);           // srcLane per lane (ds_bpermute_b32)
uint32_t sum(uint32_t v, bool pred);                    // sum over the lanes with pred (wave reduction)
uint32_t readfirst(uint32_t v);                         // lane 0's value; aborts when the lanes disagree (uniformity check)

}  // namespace simt
