// The worker slots the pipelines of a split pack share (fastore_amd/csrc/packer.h: HostGate): never more holders than slots, a class never above its cap,
// a waiting class of an earlier rank is served before a later one, everybody gets through.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include "../../fastore_amd/csrc/packer.h"

int main()
{
    const uint32_t T = 6;
    fs::HostGate g(T);
    g.rank[0] = 0; g.rank[2] = 1; g.rank[1] = 2; g.cap[0] = 4;
    std::atomic<int> held[3] = {{0}, {0}, {0}}, total{0}, bad{0}, done{0};
    std::atomic<int> firstServed[3] = {{0}, {0}, {0}};
    auto worker = [&](uint32_t cls, int tasks) {
        for (int i = 0; i < tasks; ++i) {
            g.acquire(cls);
            const int h = ++held[cls], t = ++total;
            if (t > (int)T || (cls == 0 && h > 4)) ++bad;
            std::this_thread::sleep_for(std::chrono::microseconds(200 + 50 * cls));
            --held[cls]; --total;
            g.release(cls);
            ++firstServed[cls];
        }
        ++done;
    };
    std::vector<std::thread> th;
    for (uint32_t cls = 0; cls < 3; ++cls) for (int k = 0; k < 8; ++k) th.emplace_back(worker, cls, 40);
    for (auto& t : th) t.join();
    if (bad.load() || done.load() != 24) { fprintf(stderr, "slots exceeded %d times, %d workers done\n", bad.load(), done.load()); return 1; }
    // rank order: with every slot taken by class 1 and one waiter each of classes 0 and 2, a freed slot goes to class 0, the next to class 2
    fs::HostGate q(2);
    q.rank[0] = 0; q.rank[2] = 1; q.rank[1] = 2;
    q.acquire(1); q.acquire(1);
    std::atomic<int> order{0}; int got0 = 0, got2 = 0;
    std::thread w2([&]() { q.acquire(2); got2 = ++order; q.release(2); });
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    std::thread w0([&]() { q.acquire(0); got0 = ++order; std::this_thread::sleep_for(std::chrono::milliseconds(20)); q.release(0); });
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    q.release(1);                      // one slot: class 0 first although class 2 has waited longer
    std::this_thread::sleep_for(std::chrono::milliseconds(5));
    q.release(1);
    w0.join(); w2.join();
    if (got0 != 1 || got2 != 2) { fprintf(stderr, "served in the wrong order: class 0 %d, class 2 %d\n", got0, got2); return 1; }
    printf("ok\n");
    return 0;
}
