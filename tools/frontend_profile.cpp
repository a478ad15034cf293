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
#include <time.h>
static double cpuNow() { timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: frontend_profile <bin prefix> [threads] [reps]\n"); return 2; }
    const unsigned threads = argc > 2 ? atoi(argv[2]) : 1, reps = argc > 3 ? atoi(argv[3]) : 1;
    BinFile bf; bf.open(argv[1], 256);
    ArchiveParams arch; arch.cfg = bf.config(); arch.head = bf.headData(); arch.qvz = bf.qvz();
    PackParams par; par.minBinSize = 256; par.extraReduceHardReads = true; par.minConsensusSize = 10; par.maxHammingDistance = 8; par.maxLzWindowSize = 1024; par.maxPairLzWindowSize = 1024;
    const auto& sigs = bf.stdSignatures();
    // placed unpack, as the product does it: one data batch with every bin at its known offsets, one graph batch per bin
    const size_t nb = sigs.size();
    std::vector<uint64_t> seqBase(nb + 1, 0), headBase(nb + 1, 0), recBase(nb + 1, 0);
    for (size_t k = 0; k < nb; ++k) { const BinInfo& bi = bf.bins().at(sigs[k]); seqBase[k + 1] = seqBase[k] + bi.totalRawDnaSize; headBase[k + 1] = headBase[k] + bi.totalRawHeadSize; recBase[k + 1] = recBase[k] + bi.totalRecordsCount; }
    Batch data; data.seq.resize(seqBase[nb]); data.qua.resize(seqBase[nb]); data.head.resize(headBase[nb]); data.recs.resize(recBase[nb]);
    std::vector<Batch> bins(nb);
    double t0 = 0, t1 = 0;
    for (unsigned rep = 0; rep < reps; ++rep) {
        for (auto& g : bins) g = Batch();
        t0 = now();
        std::atomic<size_t> next(0); std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t) pool.emplace_back([&]() { for (;;) { size_t i = next.fetch_add(1); if (i >= nb) break; bf.unpackPlaced(sigs[i], data, seqBase[i], headBase[i], (uint32_t)recBase[i], bins[i]); } });
        for (auto& th : pool) th.join();
        t1 = now();
        printf("unpack (placed): %zu bins, %llu records, %.1f ms on %u threads\n", nb, (unsigned long long)recBase[nb], t1 - t0, threads);
    }
    const uint64_t recs = recBase[nb];
    std::vector<BinStreams> st(sigs.size());
    for (unsigned r = 0; r < reps; ++r) {
        t1 = now(); const double c1 = cpuNow();
        std::atomic<size_t> next(0); std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t) pool.emplace_back([&]() { BinEncoder enc(par); for (;;) { size_t i = next.fetch_add(1); if (i >= sigs.size()) break; enc.encodeLz(data, bins[i], bins[i].bins[0], arch, st[i]); } });
        for (auto& th : pool) th.join();
        const double t2 = now();
        uint64_t bytes = 0; for (auto& s : st) for (auto& v : s.s) bytes += v.size();
        printf("front end: %.1f ms on %u threads, cpu %.1f ms (%.2f us per record), %.1f MB of streams\n", t2 - t1, threads, cpuNow() - c1, (cpuNow() - c1) * 1e3 / recs, bytes / 1e6);
    }
    return 0;
}
