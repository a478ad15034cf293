// Wave-execution model shared by the entropy-coder cores (ppmd_core.h, rc_core.h).
//
// On the device one entropy-coded stream is driven by ONE 64-lane wavefront.  Control flow is
// wave-uniform: every lane executes the same branches, scalar state lives in SGPR-friendly
// values (loads are made uniform with readfirstlane), and the O(n) inner steps -- table
// fills, unit copies, symbol scans, frequency prefix sums -- are spread over the lanes.
//
// The same cores are compiled for the host (one "lane") for two purposes only:
//   * the merged small-bins / N block ("block 0"), a single 100 MB-scale serial PPMd stream per
//     archive that the design deliberately keeps on a host core;
//   * a test-only emulation library (tests/emu) used to debug archive parity where no GPU is
//     present.  The product never routes a standard bin through the host build.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIP_DEVICE_COMPILE__)
  // explicit address spaces: without them the heap pointer kept in a struct degrades to FLAT accesses
  #define FS_GLOBAL __attribute__((address_space(1)))
  #define FS_LDS __attribute__((address_space(3)))
  #define FS_DEV __device__ __forceinline__
  #define FS_DEV_M __device__ __forceinline__        // member functions
  // everything is inlined into the kernel: an out-of-line callee takes the coder state by reference, which pins that
  // whole struct in scratch memory and turns every field access of the hot loop into a vector-memory instruction
  #if defined(FS_KEEP_NOINLINE)
    #define FS_DEV_NOINLINE __device__ __noinline__
  #else
    #define FS_DEV_NOINLINE __device__ __forceinline__
  #endif
  #define FS_WAVE 64
  #define FS_WIDE 1                                  // the 64-lane code paths (this build and the lock-step test emulation)
  #define FS_LANE() ((int)(threadIdx.x & 63))
  // make a loaded value wave-uniform (it already is by construction; this moves it to an SGPR)
  #define FS_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
  // a wave-uniform condition, said so (a branch the compiler takes for divergent is compiled with exec masking and
  // drags every value merged behind it into vector registers).  Via ballot: the result is a scalar mask compare, where
  // readfirstlane of the boolean would cost a round trip through a vector register.
  #define FS_UB(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)
  // order this wave's cooperative memory phase against the code that follows it.  The lanes of a wavefront execute in
  // step and its LDS and vector-memory operations are processed in issue order, so this is a compiler-level fence; it must
  // NOT be a workgroup barrier: in the two-wave form of the PPMd kernel the other wave of the workgroup never joins it
  #define FS_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
  #define FS_EMU_MEET() ((void)0)
#elif defined(FS_SIMT_EMU)
  // TEST-ONLY: the 64-lane code paths on the host, 64 fibers in lock step (tests/emu/simt.h).  Never in the product.
  #include "../../tests/emu/simt.h"
  #define FS_GLOBAL
  #define FS_LDS
  #define FS_DEV static inline
  #define FS_DEV_M inline
  #define FS_DEV_NOINLINE static
  #define FS_WAVE 64
  #define FS_WIDE 1
  #define FS_LANE() (simt::lane())
  #if defined(FS_SIMT_CHECK_UNIFORM)
    #define FS_UNI(x) (simt::readfirst((uint32_t)(x)))       // slow: every assertion is a meeting point, and is checked
  #else
    #define FS_UNI(x) ((uint32_t)(x))
  #endif
  #define FS_UB(c) (c)
  #define FS_WAVE_SYNC() simt::barrier()
  // a store whose lanes write different data is a meeting point here (on the device the lanes execute the instruction
  // together, so a later load by any lane sees all of it)
  #define FS_EMU_MEET() simt::barrier()
#else
  #define FS_GLOBAL
  #define FS_LDS
  #define FS_DEV static inline
  #define FS_DEV_M inline
  #define FS_DEV_NOINLINE static
  #define FS_WAVE 1
  #define FS_WIDE 0
  #define FS_LANE() 0
  #define FS_UNI(x) ((uint32_t)(x))
  #define FS_UB(c) (c)
  #define FS_WAVE_SYNC() ((void)0)
  #define FS_EMU_MEET() ((void)0)
#endif

// host-only event counters for design studies (tools/ppmd_paths.cpp); compiled out everywhere else
#if defined(FS_COUNTERS) && !defined(__HIP_DEVICE_COMPILE__)
  #define FS_CNT(x) (++(x), ++g_region_ops[g_region])
  struct FsRegion { int prev; explicit FsRegion(int r) : prev(g_region) { g_region = r; ++g_region_calls[r]; } ~FsRegion() { g_region = prev; } };
  #define FS_REGION(id) FsRegion fs_region_guard_(id)
  #define FS_PATH(x) (++(x))
#else
  #define FS_PATH(x) ((void)0)
  #define FS_CNT(x) ((void)0)
  #define FS_REGION(id) ((void)0)
#endif

#if !defined(FS_SYMHOOK)
  #define FS_SYMHOOK(firstCtx, lastCtx, rec, coder, succ) ((void)0)      // design-study hook (tools/ppmd_windows.cpp)
#endif

typedef FS_GLOBAL uint8_t* fs_gptr;                 // device: global address space; host: plain pointer
typedef const FS_GLOBAL uint8_t* fs_cgptr;
typedef const FS_GLOBAL uint16_t* fs_cgptr16;
typedef FS_GLOBAL uint16_t* fs_gptr16;
typedef const FS_GLOBAL uint32_t* fs_cgptr32;
typedef FS_GLOBAL uint32_t* fs_gptr32;

// ---- uniform little-endian accessors on a byte heap (2-byte aligned addresses) ----
// (FS_EMU_MEET after a load: in the lock-step test emulation every lane has read before any lane goes on to write the
// same place -- what the device gives by executing the load for all lanes at once; nothing on the device)
FS_DEV uint32_t fs_ld8(fs_cgptr p) { FS_CNT(g_ld[3]); const uint32_t v = *p; FS_EMU_MEET(); return FS_UNI(v); }
FS_DEV uint32_t fs_ld16(fs_cgptr p) { FS_CNT(g_ld[4]); const uint32_t v = *(fs_cgptr16)p; FS_EMU_MEET(); return FS_UNI(v); }
FS_DEV uint32_t fs_ld32h(fs_cgptr p)   // 32-bit value at a 2-byte aligned address
{ FS_CNT(g_ld[6]); const uint32_t v = (uint32_t)((fs_cgptr16)p)[0] | ((uint32_t)((fs_cgptr16)p)[1] << 16); FS_EMU_MEET(); return FS_UNI(v); }
FS_DEV uint32_t fs_ld32(fs_cgptr p) { FS_CNT(g_ld[5]); const uint32_t v = *(fs_cgptr32)p; FS_EMU_MEET(); return FS_UNI(v); }
// wave-uniform read of a word of the wave's LDS state that the wave also rewrites (adaptive tables, counters)
#define FS_LDS_RD(x) fs_lds_rd((uint32_t)(x))
FS_DEV uint32_t fs_lds_rd(uint32_t v) { FS_EMU_MEET(); return FS_UNI(v); }
// phase clocks of the windowed hit path (tools/ppmd_microbench.py with an -DFS_WIN_PROFILE build); nothing otherwise
#if defined(FS_WIN_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
  #define FS_PROF_NOW() ((uint64_t)__builtin_amdgcn_s_memtime())
  #define FS_PROF_ACC(w, t0) do { const uint64_t t1_ = FS_PROF_NOW(); (w) = FS_UNI(w) + (uint32_t)((t1_ - (t0)) >> 6); (t0) = t1_; } while (0)
#else
  #define FS_PROF_NOW() 0ull
  #define FS_PROF_ACC(w, t0) ((void)(t0))
#endif
// keep a loaded value alive up to this point without using it (input read-ahead of the windowed PPMd path)
#if defined(__HIP_DEVICE_COMPILE__)
  #define FS_KEEP(v) asm volatile("" :: "v"(v))
#else
  #define FS_KEEP(v) ((void)(v))
#endif
// minimum into a word of the wave's LDS state from the lanes that have something to report
#if defined(__HIP_DEVICE_COMPILE__)
  #define FS_LDS_MIN(w, v) ((void)__hip_atomic_fetch_min(&(w), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#else
  #define FS_LDS_MIN(w, v) do { if ((uint32_t)(v) < (w)) (w) = (uint32_t)(v); } while (0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
  #define FS_LDS_MAX(w, v) ((void)__hip_atomic_fetch_max(&(w), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
  #define FS_LDS_OR(w, v) ((void)__hip_atomic_fetch_or(&(w), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#else
  #define FS_LDS_MAX(w, v) do { if ((uint32_t)(v) > (w)) (w) = (uint32_t)(v); } while (0)
  #define FS_LDS_OR(w, v) do { (w) |= (uint32_t)(v); } while (0)
#endif
// statistics counter in LDS: every lane stores the same sum on the device; one lane counts in the emulation
#if defined(FS_SIMT_EMU)
  #define FS_STAT_ADD(w, x) do { if (FS_LANE() == 0) (w) += (x); } while (0)
#else
  #define FS_STAT_ADD(w, x) do { (w) = FS_UNI(w) + (x); } while (0)
#endif
// Uniform stores are executed by all lanes (same address, same value).  Issuing them from one lane was
// measured: the same number of L1->L2 write requests and the same run time, but a divergent `if` per store, which
// makes the compiler structurize the surrounding uniform control flow with exec masks (+14 % code).
FS_DEV void fs_st8(fs_gptr p, uint32_t v) { FS_CNT(g_st); *p = (uint8_t)v; }
FS_DEV void fs_st16(fs_gptr p, uint32_t v) { FS_CNT(g_st); *(fs_gptr16)p = (uint16_t)v; }
FS_DEV void fs_st32(fs_gptr p, uint32_t v) { FS_CNT(g_st); *(fs_gptr32)p = v; }
// 32-bit value at a 2-byte aligned address: lanes 0 and 1 store one half each (one instruction)
FS_DEV void fs_st32h(fs_gptr p, uint32_t v)
{
    FS_CNT(g_st);
#if FS_WIDE && !defined(FS_NO_ST48)
    const uint32_t l = (uint32_t)FS_LANE();
    if (l < 2u) ((fs_gptr16)p)[l] = (uint16_t)(l ? v >> 16 : v);
    FS_EMU_MEET();
#else
    ((fs_gptr16)p)[0] = (uint16_t)v; ((fs_gptr16)p)[1] = (uint16_t)(v >> 16);
#endif
}
// three consecutive 16-bit words (a 6-byte PPMd state) in one instruction: lane i stores word i
FS_DEV void fs_st48(fs_gptr p, uint32_t w0, uint32_t w1, uint32_t w2)
{
    FS_CNT(g_st);
#if FS_WIDE && !defined(FS_NO_ST48)
    const uint32_t l = (uint32_t)FS_LANE();
    if (l < 3u) ((fs_gptr16)p)[l] = (uint16_t)(l == 0u ? w0 : (l == 1u ? w1 : w2));
    FS_EMU_MEET();
#else
    ((fs_gptr16)p)[0] = (uint16_t)w0; ((fs_gptr16)p)[1] = (uint16_t)w1; ((fs_gptr16)p)[2] = (uint16_t)w2;
#endif
}

// ---- cross-lane helpers (host build: one lane) ----
#if defined(__HIP_DEVICE_COMPILE__)
FS_DEV uint64_t fs_ballot(bool p) { return __ballot(p); }
FS_DEV uint32_t fs_popc64(uint64_t x) { return (uint32_t)__popcll(x); }
FS_DEV uint32_t fs_ctz64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }
FS_DEV uint32_t fs_readlane(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
// value of `v` in lane `src` (per-lane source, any lane): ds_bpermute_b32
FS_DEV uint32_t fs_bperm(uint32_t v, uint32_t src) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v); }
#elif defined(FS_SIMT_EMU)
FS_DEV uint64_t fs_ballot(bool p) { return simt::ballot(p); }
FS_DEV uint32_t fs_popc64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }
FS_DEV uint32_t fs_ctz64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }
FS_DEV uint32_t fs_readlane(uint32_t v, uint32_t lane) { return simt::readlane(v, lane); }
FS_DEV uint32_t fs_bperm(uint32_t v, uint32_t src) { return simt::bperm(v, src); }
#else
FS_DEV uint64_t fs_ballot(bool p) { return p ? 1u : 0u; }
FS_DEV uint32_t fs_popc64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }
FS_DEV uint32_t fs_ctz64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }
FS_DEV uint32_t fs_readlane(uint32_t v, uint32_t) { return v; }
FS_DEV uint32_t fs_bperm(uint32_t v, uint32_t) { return v; }
#endif
// sum over the lanes with `pred` of a small value per lane: the DPP wave scan (row_shr 1/2/4/8 inside each row of 16,
// then row_bcast:15 and row_bcast:31 carry the row totals on), total read from lane 63 -- 6 adds instead of the 8
// ballot + popcount rounds of a bit-sliced sum
FS_DEV uint32_t fs_wave_sum8(uint32_t v, bool pred)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int x = pred ? (int)v : 0;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
#elif defined(FS_SIMT_EMU)
    return simt::sum(v, pred);
#else
    return pred ? v : 0u;
#endif
}

// cooperative copy / fill of 4-byte aligned regions (n bytes, n % 4 == 0), non-overlapping
FS_DEV void fs_wave_copy4(fs_gptr d, fs_cgptr s, uint32_t n)
{
    for (uint32_t i = 4u * (uint32_t)FS_LANE(); i < n; i += 4u * FS_WAVE) *(fs_gptr32)(d + i) = *(fs_cgptr32)(s + i);
    FS_WAVE_SYNC();
}
