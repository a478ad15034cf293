// How long does a process that starts right behind another wait for its device memory?  hip_alloc_probe <total GB> <pieces> [hold ms]: allocates
// total/pieces GB pieces one after the other, prints each hipMalloc's time, touches the first piece, holds everything for `hold` ms and exits
// (tools/alloc_probe.sh runs it behind itself and behind fastore_pack e processes).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <vector>
static double nowMs() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6; }
int main(int argc, char** argv)
{
    const double gb = argc > 1 ? atof(argv[1]) : 17.8; const int pieces = argc > 2 ? atoi(argv[2]) : 1; const int hold = argc > 3 ? atoi(argv[3]) : 300;
    const double t0 = nowMs();
    if (hipSetDevice(0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    (void)hipFree(nullptr);
    const double t1 = nowMs();
    std::vector<void*> p(pieces, nullptr); double worst = 0, sum = 0; int worstAt = 0;
    const size_t bytes = (size_t)(gb * 1e9 / pieces);
    for (int i = 0; i < pieces; ++i) {
        const double a = nowMs();
        if (hipMalloc(&p[i], bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
        const double d = nowMs() - a; sum += d; if (d > worst) { worst = d; worstAt = i; }
    }
    (void)hipMemset(p[0], 0, 1 << 20); (void)hipDeviceSynchronize();
    printf("%.1f GB in %d pieces: runtime start %.0f ms, allocations %.1f ms in all, the slowest %.1f ms (piece %d)\n", gb, pieces, t1 - t0, sum, worst, worstAt);
    timespec ts = {hold / 1000, (hold % 1000) * 1000000L}; nanosleep(&ts, nullptr);
    return 0;      // (no hipFree: the process's end gives the memory back, as a killed or a fast-exit process does)
}
