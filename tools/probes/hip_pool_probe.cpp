// Can a large arena pool be brought up in the background while the device already works?  (tools/gpu_r2ll.sh)
// mode d: dirty 50 GB and leave (what a previous fastore_pack process leaves behind)
// mode m: chunks by hipMalloc on a second thread, main thread keeps launching small kernels and copies
// mode v: the same through the virtual-memory calls (one reserved range, chunks mapped into it one after the other)
// mode 1: one hipMalloc of the whole pool (today's start-up), for the same clocks
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <atomic>
#include <thread>
#include <vector>
static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6; }
__global__ void touch(uint32_t* p, uint32_t n) { for (uint32_t i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += gridDim.x * blockDim.x) p[i] = i; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); fflush(stdout); exit(1); } } while (0)
int main(int argc, char** argv)
{
    const char mode = argc > 1 ? argv[1][0] : 'm';
    const int nChunks = argc > 2 ? atoi(argv[2]) : 8;
    const size_t total = (argc > 3 ? (size_t)atoi(argv[3]) : 50ull) << 30, chunk = total / nChunks;
    const double t00 = now();
    CK(hipSetDevice(0));
    printf("mode %c: runtime start-up %.1f ms\n", mode, now() - t00);
    if (mode == 'd') {
        void* p; double t0 = now(); CK(hipMalloc(&p, total)); printf("hipMalloc 50 GB %.1f ms\n", now() - t0);
        t0 = now(); CK(hipMemset(p, 0x5a, total)); CK(hipDeviceSynchronize()); printf("memset 50 GB %.1f ms\n", now() - t0);
        return 0;
    }
    if (mode == '1') {
        void* p; double t0 = now(); CK(hipMalloc(&p, total)); printf("hipMalloc %zu GB %.1f ms\n", total >> 30, now() - t0);
        t0 = now(); CK(hipMemset(p, 0x5a, total)); CK(hipDeviceSynchronize()); printf("memset %.1f ms\n", now() - t0);
        return 0;
    }
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint32_t* work; CK(hipMalloc((void**)&work, 64 << 20));
    void* pinned; CK(hipHostMalloc(&pinned, 16 << 20));
    touch<<<256, 256, 0, s>>>(work, 16 << 20); CK(hipStreamSynchronize(s));
    std::atomic<int> done(0);
    std::vector<double> tChunk(nChunks, 0);
    std::vector<void*> parts(nChunks, nullptr);
    const double tA = now();
    std::thread bg([&]() {
        (void)hipSetDevice(0);
        if (mode == 'm') {
            for (int i = 0; i < nChunks; ++i) { CK(hipMalloc(&parts[i], chunk)); tChunk[i] = now() - tA; done++; }
        } else {
            size_t gran = 0; hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
            prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
            const size_t c = (chunk + gran - 1) / gran * gran;
            void* base = nullptr; CK(hipMemAddressReserve(&base, c * nChunks, gran, nullptr, 0));
            printf("granularity %zu, reserved %p\n", gran, base);
            for (int i = 0; i < nChunks; ++i) {
                hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, c, &prop, 0));
                CK(hipMemMap((char*)base + (size_t)i * c, c, 0, h, 0));
                hipMemAccessDesc d; memset(&d, 0, sizeof d); d.location = prop.location; d.flags = hipMemAccessFlagsProtReadWrite;
                CK(hipMemSetAccess((char*)base + (size_t)i * c, c, &d, 1));
                parts[i] = (char*)base + (size_t)i * c; tChunk[i] = now() - tA; done++;
            }
        }
    });
    // the foreground: a small kernel + a 16 MB upload per round; how long does a round take while the pool comes up?
    double worst = 0, sum = 0; int rounds = 0; double worstAt = 0;
    while (done.load() < nChunks || rounds < 50) {
        const double t0 = now();
        CK(hipMemcpyAsync(work, pinned, 16 << 20, hipMemcpyHostToDevice, s));
        touch<<<256, 256, 0, s>>>(work, 4 << 20);
        CK(hipStreamSynchronize(s));
        const double d = now() - t0;
        if (d > worst) { worst = d; worstAt = t0 - tA; }
        sum += d; ++rounds;
        if (rounds > 200000) break;
    }
    bg.join();
    printf("foreground: %d rounds, mean %.2f ms, worst %.1f ms (at %.0f ms)\n", rounds, sum / rounds, worst, worstAt);
    for (int i = 0; i < nChunks; ++i) printf("  chunk %d (%.1f GB) there at %.1f ms\n", i, chunk / 1e9, tChunk[i]);
    // the chunks are usable: write every one of them
    double t0 = now();
    for (int i = 0; i < nChunks; ++i) touch<<<1024, 256, 0, s>>>((uint32_t*)parts[i], (uint32_t)(chunk / 4 > 0xffffffffull ? 0xffffffffu : chunk / 4));
    CK(hipStreamSynchronize(s));
    printf("writing all chunks %.1f ms; since start %.1f ms\n", now() - t0, now() - t00);
    return 0;
}
