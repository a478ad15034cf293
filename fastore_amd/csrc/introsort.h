// Introsort with the exact element movements of libstdc++'s std::sort (bits/stl_algo.h:
// __introsort_loop / __unguarded_partition_pivot / __final_insertion_sort / heap fallback).
// The reference sorts reads with comparators that tie on duplicates (FastqRecord.h:226-257,
// ContigBuilder.h:36-39), so the *unstable* tie order of its std::sort is part of the archive
// bytes; this restatement makes that order independent of the C++ runtime in use and is written
// over plain index arrays so it can move to the device later.
#pragma once
#include <stddef.h>
#include <utility>

namespace fs {

template <class T, class Less> struct IntroSort {
    T* a; Less less;
    IntroSort(T* arr, Less l) : a(arr), less(l) {}

    static int lg(ptrdiff_t n) { int k = 0; while (n > 1) { n >>= 1; ++k; } return k; }

    void moveMedianToFirst(ptrdiff_t result, ptrdiff_t x, ptrdiff_t y, ptrdiff_t z)
    {
        if (less(a[x], a[y])) {
            if (less(a[y], a[z])) std::swap(a[result], a[y]);
            else if (less(a[x], a[z])) std::swap(a[result], a[z]);
            else std::swap(a[result], a[x]);
        } else if (less(a[x], a[z])) std::swap(a[result], a[x]);
        else if (less(a[y], a[z])) std::swap(a[result], a[z]);
        else std::swap(a[result], a[y]);
    }
    ptrdiff_t unguardedPartition(ptrdiff_t first, ptrdiff_t last, ptrdiff_t pivot)
    {
        for (;;) {
            while (less(a[first], a[pivot])) ++first;
            --last;
            while (less(a[pivot], a[last])) --last;
            if (!(first < last)) return first;
            std::swap(a[first], a[last]);
            ++first;
        }
    }
    void pushHeap(ptrdiff_t first, ptrdiff_t hole, ptrdiff_t top, T value)
    {
        ptrdiff_t parent = (hole - 1) / 2;
        while (hole > top && less(a[first + parent], value)) {
            a[first + hole] = std::move(a[first + parent]);
            hole = parent; parent = (hole - 1) / 2;
        }
        a[first + hole] = std::move(value);
    }
    void adjustHeap(ptrdiff_t first, ptrdiff_t hole, ptrdiff_t len, T value)
    {
        const ptrdiff_t top = hole;
        ptrdiff_t child = hole;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            if (less(a[first + child], a[first + (child - 1)])) child--;
            a[first + hole] = std::move(a[first + child]);
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            a[first + hole] = std::move(a[first + (child - 1)]);
            hole = child - 1;
        }
        pushHeap(first, hole, top, std::move(value));
    }
    void heapSort(ptrdiff_t first, ptrdiff_t last)     // __partial_sort(first, last, last)
    {
        const ptrdiff_t len = last - first;
        if (len >= 2) {
            for (ptrdiff_t parent = (len - 2) / 2;; --parent) {
                T v = std::move(a[first + parent]);
                adjustHeap(first, parent, len, std::move(v));
                if (parent == 0) break;
            }
        }
        while (last - first > 1) {
            --last;
            T v = std::move(a[last]);
            a[last] = std::move(a[first]);
            adjustHeap(first, 0, last - first, std::move(v));
        }
    }
    void introLoop(ptrdiff_t first, ptrdiff_t last, int depth)
    {
        while (last - first > 16) {
            if (depth == 0) { heapSort(first, last); return; }
            --depth;
            const ptrdiff_t mid = first + (last - first) / 2;
            moveMedianToFirst(first, first + 1, mid, last - 1);
            const ptrdiff_t cut = unguardedPartition(first + 1, last, first);
            introLoop(cut, last, depth);
            last = cut;
        }
    }
    void unguardedLinearInsert(ptrdiff_t last)
    {
        T val = std::move(a[last]);
        ptrdiff_t next = last - 1;
        while (less(val, a[next])) { a[last] = std::move(a[next]); last = next; --next; }
        a[last] = std::move(val);
    }
    void insertionSort(ptrdiff_t first, ptrdiff_t last)
    {
        if (first == last) return;
        for (ptrdiff_t i = first + 1; i != last; ++i) {
            if (less(a[i], a[first])) {
                T val = std::move(a[i]);
                for (ptrdiff_t j = i; j > first; --j) a[j] = std::move(a[j - 1]);
                a[first] = std::move(val);
            } else unguardedLinearInsert(i);
        }
    }
    void sort(ptrdiff_t n)
    {
        if (n <= 0) return;
        introLoop(0, n, lg(n) * 2);
        if (n > 16) {
            insertionSort(0, 16);
            for (ptrdiff_t i = 16; i != n; ++i) unguardedLinearInsert(i);
        } else insertionSort(0, n);
    }
};

template <class T, class Less> void introsort(T* a, size_t n, Less less) { IntroSort<T, Less>(a, less).sort((ptrdiff_t)n); }

}  // namespace fs
