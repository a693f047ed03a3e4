// tests/native/refsel_harness.cpp -- host build of csrc/refsel.h beside std::nth_element (libstdc++), for tests/test_refsel.py.
// TEST CODE: compiled by the test into a temporary shared object.
#include <algorithm>
#include <cstdint>
#include <vector>
static int g_heap_runs = 0;
#define LIO_REFSEL_ON_HEAP (g_heap_runs++)
#include "refsel.h"

namespace {
struct DistPoint {  // the shape of IVoxNode::DistPoint (ivox3d_node.hpp:71-84): ordered by dist alone
    double dist;
    uint32_t id;
    bool operator<(const DistPoint& o) const { return dist < o.dist; }
};
float as_float(uint32_t b) { float f; __builtin_memcpy(&f, &b, 4); return f; }
}  // namespace

extern "C" {
int refsel_heap_runs() { return g_heap_runs; }
// nth_element(first, nth, last) on n records: ids as std leaves them -> out_std, as refsel leaves them -> out_mine
void refsel_nth(const uint32_t* dbits, int n, int first, int nth, int last, uint32_t* out_std, uint32_t* out_mine) {
    std::vector<DistPoint> v(n);
    std::vector<lio::refsel::Rec> r(n);
    for (int i = 0; i < n; i++) { v[i] = {(double)as_float(dbits[i]), (uint32_t)i}; r[i] = {dbits[i], (uint32_t)i}; }
    std::nth_element(v.begin() + first, v.begin() + nth, v.begin() + last);
    lio::refsel::nth_element(r.data(), first, nth, last);
    for (int i = 0; i < n; i++) { out_std[i] = v[i].id; out_mine[i] = r[i].id; }
}
// a whole query: voxel sizes cnt[0..nv), candidates in order; returns the size of the list, ids as each side leaves them
int refsel_query(const uint32_t* dbits, const int* cnt, int nv, int k, uint32_t* out_std, uint32_t* out_mine) {
    std::vector<DistPoint> v;
    std::vector<lio::refsel::Rec> r;
    int at = 0, size = 0;
    for (int s = 0; s < nv; s++) {
        const size_t old = v.size();
        for (int j = 0; j < cnt[s]; j++, at++) { v.push_back({(double)as_float(dbits[at]), (uint32_t)at}); r.push_back({dbits[at], (uint32_t)at}); }
        if (old + k >= v.size()) {
        } else {
            std::nth_element(v.begin() + old, v.begin() + old + k - 1, v.end());
            v.resize(old + k);
        }
        size = lio::refsel::voxel_cut(r.data(), size, (int)r.size(), k);
        r.resize(size);
    }
    if ((int)v.size() > k) { std::nth_element(v.begin(), v.begin() + k - 1, v.end()); v.resize(k); }
    if (!v.empty()) std::nth_element(v.begin(), v.begin(), v.end());
    size = r.empty() ? 0 : lio::refsel::final_cut(r.data(), size, k);
    for (size_t i = 0; i < v.size(); i++) out_std[i] = v[i].id;
    for (int i = 0; i < size; i++) out_mine[i] = r[i].id;
    return (int)v.size() == size ? size : -1;
}
}
