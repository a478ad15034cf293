// Micro-probe (design study): does a store in front of a dependent load delay it?  One wave, wave-uniform code like
// the coder's: a pointer chase over a small (L2-resident) table, with and without a store per step.
//   hipcc --offload-arch=gfx950 -O3 -o build/store_ack_probe tools/probes/store_ack_probe.hip && build/store_ack_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

template <int MODE>
__global__ void chase(unsigned* next, unsigned* sink, unsigned sinkMask, int steps, unsigned long long* cycles, unsigned* result)
{
    unsigned p = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < steps; ++i) {
        if (MODE == 1) sink[((p * 2654435761u) & sinkMask) * 16u] = (unsigned)i;                 // store first, then the dependent load
        unsigned q = next[p];
        if (MODE == 2) sink[((p * 2654435761u) & sinkMask) * 16u] = (unsigned)i;                 // load issued first, store behind it
        if (MODE == 3) { if (threadIdx.x == 0) sink[((p * 2654435761u) & sinkMask) * 16u] = (unsigned)i; }   // one-lane store first
        p = (unsigned)__builtin_amdgcn_readfirstlane((int)q);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { *cycles = t1 - t0; *result = p; }
}

static void run(int N, unsigned sinkLines, int steps);
int main()
{
    run(1 << 14, 1 << 10, 20000);            // 64 KB table, 64 KB of store lines: everything L2 resident
    run(1 << 26, 1 << 24, 20000);            // 256 MB table, 1 GB of store lines: misses all the way to HBM
    return 0;
}
static void run(int N, unsigned sinkLines, int steps)
{
    std::vector<unsigned> perm(N); std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(5); std::shuffle(perm.begin() + 1, perm.end(), rng);
    std::vector<unsigned> next(N);
    for (int i = 0; i < N; ++i) next[perm[i]] = perm[(i + 1) % N];
    unsigned *dNext, *dSink, *dRes; unsigned long long* dCyc;
    hipMalloc(&dNext, N * 4); hipMalloc(&dSink, (size_t)sinkLines * 64 + 64); hipMalloc(&dRes, 4); hipMalloc(&dCyc, 8);
    hipMemcpy(dNext, next.data(), N * 4, hipMemcpyHostToDevice);
    const char* names[] = {"loads only", "store then load", "load then store", "one-lane store then load"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            switch (mode) {
            case 0: hipLaunchKernelGGL(chase<0>, dim3(1), dim3(64), 0, 0, dNext, dSink, sinkLines - 1, steps, dCyc, dRes); break;
            case 1: hipLaunchKernelGGL(chase<1>, dim3(1), dim3(64), 0, 0, dNext, dSink, sinkLines - 1, steps, dCyc, dRes); break;
            case 2: hipLaunchKernelGGL(chase<2>, dim3(1), dim3(64), 0, 0, dNext, dSink, sinkLines - 1, steps, dCyc, dRes); break;
            default: hipLaunchKernelGGL(chase<3>, dim3(1), dim3(64), 0, 0, dNext, dSink, sinkLines - 1, steps, dCyc, dRes); break;
            }
            hipDeviceSynchronize();
            unsigned long long c = 0; hipMemcpy(&c, dCyc, 8, hipMemcpyDeviceToHost);
            if (rep) printf("table %4d MB, stores over %4u MB: %-28s %8.1f cycles per step\n", N / 262144, sinkLines / 16384, names[mode], (double)c / steps);
        }
    hipFree(dNext); hipFree(dSink); hipFree(dRes); hipFree(dCyc);
}
