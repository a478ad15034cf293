// Where does a one-shot process spend its HIP start-up and teardown?  (tools/gpu_r2jj.sh)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <thread>
#include <vector>
#include <unistd.h>
static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6; }
__global__ void touch(uint32_t* p) { p[threadIdx.x] = threadIdx.x; }
#define T(label, expr) do { double t0 = now(); hipError_t e = (expr); printf("%-44s %8.1f ms %s\n", label, now() - t0, e == hipSuccess ? "" : hipGetErrorString(e)); } while (0)
int main(int argc, char** argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const size_t GB = 1ull << 30;
    const double t00 = now();
    T("hipSetDevice(0) [runtime start-up]", hipSetDevice(0));
    hipDeviceProp_t prop; T("hipGetDeviceProperties", hipGetDeviceProperties(&prop, 0));
    size_t fr, to; T("hipMemGetInfo", hipMemGetInfo(&fr, &to));
    void* big = nullptr; std::vector<void*> parts(8, nullptr);
    if (mode == 0) { T("hipMalloc 50 GB", hipMalloc(&big, 50 * GB)); }
    else if (mode == 1) { for (int i = 0; i < 8; ++i) T("hipMalloc 6.25 GB", hipMalloc(&parts[i], 50 * GB / 8)); }
    else if (mode == 2) {
        double t0 = now(); std::vector<std::thread> th;
        for (int i = 0; i < 8; ++i) th.emplace_back([&, i]() { (void)hipSetDevice(0); (void)hipMalloc(&parts[i], 50 * GB / 8); });
        for (auto& t : th) t.join();
        printf("%-44s %8.1f ms\n", "8 threads x hipMalloc 6.25 GB", now() - t0);
    } else if (mode == 3) { T("hipMalloc 16 GB", hipMalloc(&big, 16 * GB)); }
    else if (mode == 4) { T("hipExtMallocWithFlags 50 GB uncached", hipExtMallocWithFlags(&big, 50 * GB, hipDeviceMallocUncached)); }
    void* small = nullptr; T("hipMalloc 64 MB", hipMalloc(&small, 64 << 20));
    void* small2 = nullptr; T("hipMalloc 1 GB", hipMalloc(&small2, 1 * GB));
    void* hst = nullptr; T("hipHostMalloc 256 MB", hipHostMalloc(&hst, 256 << 20));
    void* hst2 = nullptr; T("hipHostMalloc 256 MB (2nd)", hipHostMalloc(&hst2, 256 << 20));
    { double t0 = now(); void* m = malloc(256 << 20); memset(m, 1, 256 << 20); printf("%-44s %8.1f ms\n", "malloc+memset 256 MB", now() - t0);
      T("hipHostRegister 256 MB", hipHostRegister(m, 256 << 20, hipHostRegisterDefault)); }
    hipStream_t s; T("hipStreamCreate", hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    { double t0 = now(); touch<<<1, 64, 0, s>>>((uint32_t*)small); hipError_t e = hipStreamSynchronize(s); printf("%-44s %8.1f ms %s\n", "first kernel launch + sync [code object]", now() - t0, e == hipSuccess ? "" : hipGetErrorString(e)); }
    { double t0 = now(); touch<<<1, 64, 0, s>>>((uint32_t*)small); (void)hipStreamSynchronize(s); printf("%-44s %8.1f ms\n", "second launch + sync", now() - t0); }
    T("hipMemcpy H2D 256 MB pinned", hipMemcpy(small2, hst, 256 << 20, hipMemcpyHostToDevice));
    printf("%-44s %8.1f ms\n", "since start", now() - t00);
    if (argc > 2 && atoi(argv[2]) == 1) {
        if (big) T("hipFree big", hipFree(big));
        for (void* p : parts) if (p) T("hipFree part", hipFree(p));
        T("hipFree 1 GB", hipFree(small2));
        T("hipHostFree 256 MB", hipHostFree(hst));
        printf("%-44s %8.1f ms\n", "since start (after frees)", now() - t00);
    }
    fflush(stdout);
    if (argc > 2 && atoi(argv[2]) == 2) _exit(0);
    return 0;
}
