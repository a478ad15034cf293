// Probe: what one wavefront alone pays per instruction in dependent scalar chains, VALU->SALU hand-offs and branches
// (the shape of the range-coder loop of ppmd_window.h).  hipcc --offload-arch=gfx950 -O2 -o build/salu_probe tools/probes/salu_latency_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
__global__ void probe(uint64_t* out, uint32_t seed)
{
    uint32_t a = seed, b = seed | 1u, c = 0;
    uint64_t t0, t1;
    // 0: dependent s_add_u32 chain
    t0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    asm volatile(REP64("s_add_u32 %0, %0, %1\n") : "+s"(a) : "s"(b));
    t1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    if (threadIdx.x == 0) out[0] = t1 - t0;
    // 1: dependent s_mul_i32 chain
    t0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    asm volatile(REP64("s_mul_i32 %0, %0, %1\n") : "+s"(a) : "s"(b));
    t1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    if (threadIdx.x == 0) out[1] = t1 - t0;
    // 2: dependent s_mul_hi_u32 chain
    t0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    asm volatile(REP64("s_mul_hi_u32 %0, %0, %1\n") : "+s"(a) : "s"(b));
    t1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    if (threadIdx.x == 0) out[2] = t1 - t0;
    // 3: independent s_add (two chains interleaved)
    t0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    asm volatile(REP64("s_add_u32 %0, %0, %2\ns_add_u32 %1, %1, %2\n") : "+s"(a), "+s"(c) : "s"(b));
    t1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    if (threadIdx.x == 0) out[3] = t1 - t0;
    // 4: v_readlane -> s_add (VALU writes SGPR, SALU consumes), chain through the SALU result as lane index
    uint32_t v = threadIdx.x * 7u + seed;
    uint32_t idx = seed & 63u;
    t0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    asm volatile(REP64("v_readlane_b32 %0, %2, %1\ns_and_b32 %1, %0, 63\n") : "+s"(a), "+s"(idx) : "v"(v));
    t1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    if (threadIdx.x == 0) out[4] = t1 - t0;
    // 5: dependent v_add chain
    t0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(v) : "v"(b));
    t1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    if (threadIdx.x == 0) out[5] = t1 - t0;
    // 6: taken branches (64 jumps to the next instruction group)
    t0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    asm volatile(REP64("s_branch 1f\ns_nop 0\n1:\n"));
    t1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    if (threadIdx.x == 0) out[6] = t1 - t0;
    // 7: not-taken conditional branches
    t0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    asm volatile("s_cmp_eq_u32 0, 1\n" REP64("s_cbranch_scc1 1f\n1:\n"));
    t1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    if (threadIdx.x == 0) out[7] = t1 - t0;
    // 8: s_cmp + not-taken branch after a dependent op (the coder's test)
    t0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    asm volatile(REP64("s_add_u32 %0, %0, %1\ns_cmp_eq_u32 %0, 12345\ns_cbranch_scc1 1f\n1:\n") : "+s"(a) : "s"(b));
    t1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    if (threadIdx.x == 0) out[8] = t1 - t0;
    // 9: v_cmp -> v_cndmask via SGPR pair (the select chains of the old round code)
    uint32_t w = threadIdx.x;
    t0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    asm volatile(REP64("v_cmp_eq_u32 vcc, %0, %1\nv_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(w) : "v"(v) : "vcc");
    t1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    if (threadIdx.x == 0) out[9] = t1 - t0;
    // 10: ds_bpermute dependent chain
    t0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    asm volatile(REP8("ds_bpermute_b32 %0, %1, %0\ns_waitcnt lgkmcnt(0)\n") : "+v"(w) : "v"(v));
    t1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)");
    if (threadIdx.x == 0) out[10] = t1 - t0;
    if (threadIdx.x == 0) out[15] = a + c + idx;
    if (v == 0x12345 || w == 0x54321) out[14] = v + w;
}
int main()
{
    uint64_t* d; hipMalloc(&d, 128); hipMemset(d, 0, 128);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 12345u); hipDeviceSynchronize(); }
    uint64_t h[16]; hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    const char* names[] = {"dependent s_add_u32", "dependent s_mul_i32", "dependent s_mul_hi_u32", "two independent s_add chains (per pair)", "v_readlane -> s_and -> (lane index) pair",
                           "dependent v_add_u32", "taken s_branch (+s_nop)", "not-taken s_cbranch", "s_add + s_cmp + not-taken branch (triple)", "v_cmp -> v_cndmask via vcc (pair)", "ds_bpermute + wait (x8)"};
    const int reps[] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 8};
    for (int i = 0; i < 11; ++i) printf("%-48s %6.1f clocks each (%llu total)\n", names[i], (double)h[i] / reps[i], (unsigned long long)h[i]);
    return 0;
}
