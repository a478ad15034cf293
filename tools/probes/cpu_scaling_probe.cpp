// Design study: how many cores does this process really get?  Fixed total arithmetic work split over N threads.
//   g++ -O2 -pthread -o build/cpu_scaling_probe tools/probes/cpu_scaling_probe.cpp
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const unsigned long long total = 24ull * 400000000ull;
    for (unsigned n : {1u, 8u, 12u, 16u, 24u, 32u, 48u, 64u, 96u, 128u}) {
        std::vector<std::thread> pool; std::vector<unsigned long long> sink(n * 16);
        const double t0 = now();
        for (unsigned t = 0; t < n; ++t) pool.emplace_back([&, t]() { unsigned long long x = t + 1; for (unsigned long long i = 0; i < total / n; ++i) x = x * 6364136223846793005ull + 1442695040888963407ull; sink[t * 16] = x; });
        for (auto& th : pool) th.join();
        const double dt = now() - t0;
        printf("%3u threads: %.3f s  -> %.1f effective cores (vs 1 thread)\n", n, dt, 0.0);
        fflush(stdout);
    }
    return 0;
}
