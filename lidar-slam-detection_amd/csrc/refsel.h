// refsel.h -- the selection the reference's neighbour search runs on its candidate list, restated for the device.
//
// IVox::GetClosestPoint (/root/reference/slam/mapping/fastlio/include/ivox3d/ivox3d.h:139-171) and IVoxNode::KNNPointByCondition
// (ivox3d_node.hpp:107-127) pick the five nearest candidates with std::nth_element over DistPoint records compared by `dist` alone.
// Where two candidates are exactly as far as the fifth nearest, which of them stays is decided by what that function does to the
// particular sequence it is given -- candidates in stencil order, inside a voxel in push_back order, each voxel cut to five by its own
// nth_element first.  std::nth_element is not part of the reference tree: it is libstdc++'s (the reference is built with GCC; pinned here
// to GCC 11.4's bits/stl_algo.h, unchanged in this function since 4.x): introselect = median-of-three quick-select down to ranges of three
// elements, finished by an insertion sort, with a heap-select fallback after 2 * floor(log2(n)) partitions.  The statements below restate
// that published algorithm over an array of (distance bits, point id) pairs; tests/test_refsel.py runs them on the host against
// std::nth_element itself (random and adversarial sequences, permutation compared element by element), tests/test_gpu_parity.py on the
// device against the reference's own ivox3d.h compiled here.
//
// Distances are f32 squared norms >= +0: the order of their bit patterns as unsigned integers is their numeric order (the reference
// compares the same values widened to double).
#pragma once
#include <stdint.h>

#ifndef LIO_HD
#ifdef __HIPCC__
#define LIO_HD __host__ __device__
#else
#define LIO_HD
#endif
#endif

#ifndef LIO_REFSEL_ON_HEAP
#define LIO_REFSEL_ON_HEAP  // (a test hook: counts the runs that reach the heap-select fallback)
#endif

namespace lio {
namespace refsel {

struct Rec {
    uint32_t d;   // bits of the squared distance
    uint32_t id;  // what travels with it (pool index)
};

LIO_HD inline bool less(const Rec& a, const Rec& b) { return a.d < b.d; }
LIO_HD inline void swap_rec(Rec* a, int i, int j) { const Rec t = a[i]; a[i] = a[j]; a[j] = t; }

// sift `value` up from `hole` towards `top`
LIO_HD inline void push_heap(Rec* a, int first, int hole, int top, Rec value) {
    int parent = (hole - 1) / 2;
    while (hole > top && less(a[first + parent], value)) {
        a[first + hole] = a[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[first + hole] = value;
}
// sink the hole to a leaf along the larger children, then sift `value` up from there
LIO_HD inline void adjust_heap(Rec* a, int first, int hole, int len, Rec value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (less(a[first + child], a[first + child - 1])) child--;
        a[first + hole] = a[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[first + hole] = a[first + child - 1];
        hole = child - 1;
    }
    push_heap(a, first, hole, top, value);
}
// max-heap of [first, middle); every later element smaller than the heap's top replaces it
LIO_HD inline void heap_select(Rec* a, int first, int middle, int last) {
    const int len = middle - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) {
            const Rec v = a[first + parent];
            adjust_heap(a, first, parent, len, v);
            if (parent == 0) break;
            parent--;
        }
    }
    for (int i = middle; i < last; i++) {
        if (less(a[i], a[first])) {
            const Rec v = a[i];
            a[i] = a[first];
            adjust_heap(a, first, 0, len, v);
        }
    }
}
LIO_HD inline void insertion_sort(Rec* a, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; i++) {
        const Rec v = a[i];
        if (less(v, a[first])) {
            for (int k = i; k > first; k--) a[k] = a[k - 1];
            a[first] = v;
        } else {
            int at = i, next = i - 1;
            while (less(v, a[next])) {
                a[at] = a[next];
                at = next;
                next--;
            }
            a[at] = v;
        }
    }
}
// the median of a[x], a[y], a[z] goes to a[result]
LIO_HD inline void median_to_first(Rec* a, int result, int x, int y, int z) {
    if (less(a[x], a[y])) {
        if (less(a[y], a[z])) swap_rec(a, result, y);
        else if (less(a[x], a[z])) swap_rec(a, result, z);
        else swap_rec(a, result, x);
    } else if (less(a[x], a[z])) swap_rec(a, result, x);
    else if (less(a[y], a[z])) swap_rec(a, result, z);
    else swap_rec(a, result, y);
}
LIO_HD inline int partition_pivot(Rec* a, int first, int last) {
    const int mid = first + (last - first) / 2;
    median_to_first(a, first, first + 1, mid, last - 1);
    int lo = first + 1, hi = last;
    for (;;) {
        while (less(a[lo], a[first])) lo++;
        hi--;
        while (less(a[first], a[hi])) hi--;
        if (!(lo < hi)) return lo;
        swap_rec(a, lo, hi);
        lo++;
    }
}
// std::nth_element(a + first, a + nth, a + last)
LIO_HD inline void nth_element(Rec* a, int first, int nth, int last) {
    if (first == last || nth == last) return;
    int depth = 0;
    for (int n = last - first; n > 1; n >>= 1) depth++;
    depth *= 2;
    while (last - first > 3) {
        if (depth == 0) {
            LIO_REFSEL_ON_HEAP;
            heap_select(a, first, nth + 1, last);
            swap_rec(a, first, nth);
            return;
        }
        depth--;
        const int cut = partition_pivot(a, first, last);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    insertion_sort(a, first, last);
}

// One voxel's share of the candidate list (ivox3d_node.hpp:107-127): its in-range points were appended at a[old_size .. size) in push_back
// order; more than k of them are cut to the k that nth_element leaves in front.  Returns the new size.
LIO_HD inline int voxel_cut(Rec* a, int old_size, int size, int k) {
    if (old_size + k >= size) return size;
    nth_element(a, old_size, old_size + k - 1, size);
    return old_size + k;
}
// the tail of GetClosestPoint (ivox3d.h:156-164): cut the whole list to max_num, then bring the nearest to the front.  Returns the size.
LIO_HD inline int final_cut(Rec* a, int size, int max_num) {
    if (size > max_num) {
        nth_element(a, 0, max_num - 1, size);
        size = max_num;
    }
    nth_element(a, 0, 0, size);
    return size;
}

}  // namespace refsel
}  // namespace lio
