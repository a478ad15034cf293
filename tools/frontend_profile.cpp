// Design study (host only): unpack + front end of one binned library, no device work.  Times the stages per thread
// count; build with -pg for a gprof profile.
//   g++ -O2 -g -std=c++17 -pthread -Ifastore_amd/csrc -o build/frontend_profile tools/frontend_profile.cpp \
//       fastore_amd/csrc/{binfile,frontend,qvz}.cpp
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <memory>
#include <thread>
#include <vector>
#include "binfile.h"
#include "frontend.h"
using namespace fs;
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: frontend_profile <bin prefix> [threads] [reps]\n"); return 2; }
    const unsigned threads = argc > 2 ? atoi(argv[2]) : 1, reps = argc > 3 ? atoi(argv[3]) : 1;
    BinFile bf; bf.open(argv[1], 256);
    ArchiveParams arch; arch.cfg = bf.config(); arch.head = bf.headData(); arch.qvz = bf.qvz();
    PackParams par; par.minBinSize = 256; par.extraReduceHardReads = true; par.minConsensusSize = 10; par.maxHammingDistance = 8; par.maxLzWindowSize = 1024; par.maxPairLzWindowSize = 1024;
    const auto& sigs = bf.stdSignatures();
    double t0 = now();
    std::vector<Batch> bins(sigs.size());
    {
        std::atomic<size_t> next(0); std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t) pool.emplace_back([&]() { for (;;) { size_t i = next.fetch_add(1); if (i >= sigs.size()) break; bf.unpack(sigs[i], bins[i], true); } });
        for (auto& th : pool) th.join();
    }
    double t1 = now();
    uint64_t recs = 0; for (auto& b : bins) recs += b.recs.size();
    printf("unpack: %zu bins, %llu records, %.1f ms on %u threads\n", sigs.size(), (unsigned long long)recs, t1 - t0, threads);
    std::vector<BinStreams> st(sigs.size());
    for (unsigned r = 0; r < reps; ++r) {
        t1 = now();
        std::atomic<size_t> next(0); std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t) pool.emplace_back([&]() { BinEncoder enc(par); for (;;) { size_t i = next.fetch_add(1); if (i >= sigs.size()) break; enc.encodeLz(bins[i], bins[i].bins[0], arch, st[i]); } });
        for (auto& th : pool) th.join();
        const double t2 = now();
        uint64_t bytes = 0; for (auto& s : st) for (auto& v : s.s) bytes += v.size();
        printf("front end: %.1f ms on %u threads (%.2f us per record), %.1f MB of streams\n", t2 - t1, threads, (t2 - t1) * 1e3 * threads / recs, bytes / 1e6);
    }
    return 0;
}
