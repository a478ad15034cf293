// What does a process pay at exit for its HIP streams (hardware queues)?  (tools/gpu_r2xx.sh)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>
#include <vector>
static long long wallMs() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return (long long)ts.tv_sec * 1000 + ts.tv_nsec / 1000000; }
__global__ void touch(uint32_t* p) { p[threadIdx.x + blockIdx.x * blockDim.x] = threadIdx.x; }
int main(int argc, char** argv)
{
    const int nStreams = argc > 1 ? atoi(argv[1]) : 16;
    const size_t gb = argc > 2 ? atoi(argv[2]) : 0;          // device memory to allocate and write
    const size_t hostGb = argc > 3 ? atoi(argv[3]) : 0;      // host memory to touch
    (void)hipSetDevice(0);
    std::vector<hipStream_t> s(nStreams);
    uint32_t* d; (void)hipMalloc((void**)&d, 64 << 20);
    for (int i = 0; i < nStreams; ++i) { (void)hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); touch<<<64, 256, 0, s[i]>>>(d + 65536 * i); }
    for (int i = 0; i < nStreams; ++i) (void)hipStreamSynchronize(s[i]);
    if (gb) { void* p; (void)hipMalloc(&p, gb << 30); (void)hipMemset(p, 1, gb << 30); (void)hipDeviceSynchronize(); }
    if (hostGb) { char* h = (char*)malloc(hostGb << 30); for (size_t o = 0; o < (hostGb << 30); o += 4096) h[o] = 1; }
    const int nAlloc = argc > 4 ? atoi(argv[4]) : 0;        // separate device allocations of 32 MB each, written
    for (int i = 0; i < nAlloc; ++i) { void* p; (void)hipMalloc(&p, 32 << 20); (void)hipMemsetAsync(p, 1, 32 << 20, s[0]); }
    const int nPinned = argc > 5 ? atoi(argv[5]) : 0;       // separate pinned host allocations of 8 MB each
    for (int i = 0; i < nPinned; ++i) { void* p; (void)hipHostMalloc(&p, 8 << 20); }
    (void)hipDeviceSynchronize();
    printf("streams %d, device %zu GB, host %zu GB, %d device allocations, %d pinned allocations: leaving at %lld\n", nStreams, gb, hostGb, nAlloc, nPinned, wallMs());
    fflush(stdout);
    _exit(0);
}
