// TEST-ONLY: fs::introsort must move elements exactly like libstdc++'s std::sort, ties included.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../fastore_amd/csrc/introsort.h"
struct E { int key; int id; };
int main()
{
    unsigned seed = 12345; auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    int bad = 0;
    const int sizes[] = {0, 1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 100, 257, 1000, 4097, 20000};
    for (int n : sizes) for (int keys : {1, 2, 3, 7, 50, 1000000}) for (int rep = 0; rep < 3; ++rep) {
        std::vector<E> a(n); for (int i = 0; i < n; ++i) a[i] = E{(int)(rnd() % keys), i};
        if (rep == 1) std::sort(a.begin(), a.end(), [](const E& x, const E& y) { return x.key < y.key; });           // presorted
        if (rep == 2) std::sort(a.begin(), a.end(), [](const E& x, const E& y) { return x.key > y.key; });           // reversed
        std::vector<E> b = a;
        auto less = [](const E& x, const E& y) { return x.key < y.key; };
        std::sort(a.begin(), a.end(), less);
        fs::introsort(b.data(), b.size(), less);
        for (int i = 0; i < n; ++i) if (a[i].id != b[i].id) { ++bad; break; }
    }
    // adversarial: organ-pipe + many equal keys to reach the heap-sort fallback
    for (int n : {3000, 50000}) {
        std::vector<E> a(n); for (int i = 0; i < n; ++i) a[i] = E{(i < n / 2 ? i : n - i) / 3, i};
        std::vector<E> b = a; auto less = [](const E& x, const E& y) { return x.key < y.key; };
        std::sort(a.begin(), a.end(), less); fs::introsort(b.data(), b.size(), less);
        for (int i = 0; i < n; ++i) if (a[i].id != b[i].id) { ++bad; break; }
    }
    printf("%s\n", bad ? "MISMATCH" : "OK");
    return bad ? 1 : 0;
}
