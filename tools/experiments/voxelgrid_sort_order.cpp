// What pcl::VoxelGrid's own std::sort (idx-only comparator: unstable) does to the order of the addends inside a voxel, against the input order the
// oracle and the product fix: how many voxels get another order, how many centroids differ, by how many ulp.  libstdc++'s introsort, the one the
// reference's binaries run.   g++ -O2 -o /tmp/vgsort tools/experiments/voxelgrid_sort_order.cpp && /tmp/vgsort [cloud.bin leaf]
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
struct cpi { unsigned idx; unsigned cloud_point_index; bool operator<(const cpi& p) const { return idx < p.idx; } };
int main(int argc, char** argv) {
    std::mt19937 rng(5);
    std::normal_distribution<float> nx(0.f, 30.f), nz(0.f, 1.5f);
    int n = 120000; float leaf = 0.5f;
    std::vector<float> p(4 * (size_t)n);
    for (int i = 0; i < n; i++) { p[4*i] = nx(rng); p[4*i+1] = nx(rng); p[4*i+2] = nz(rng); p[4*i+3] = (float)(i % 255); }
    if (argc > 2) {  // a cloud of x y z intensity f32 records (KITTI .bin layout) and a leaf size
        FILE* f = fopen(argv[1], "rb");
        if (!f) return 1;
        fseek(f, 0, SEEK_END); n = (int)(ftell(f) / 16); fseek(f, 0, SEEK_SET);
        p.resize(4 * (size_t)n);
        if (fread(p.data(), 16, n, f) != (size_t)n) return 1;
        fclose(f);
        leaf = (float)atof(argv[2]);
    }
    const float inv = 1.0f / leaf;
    float mn[3] = {1e30f,1e30f,1e30f}, mx[3] = {-1e30f,-1e30f,-1e30f};
    for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], p[4*i+k]); mx[k] = std::max(mx[k], p[4*i+k]); }
    int minb[3], divb[3];
    for (int k = 0; k < 3; k++) { minb[k] = (int)std::floor(mn[k] * inv); divb[k] = (int)std::floor(mx[k] * inv) - minb[k] + 1; }
    std::vector<cpi> a(n), b;
    for (int i = 0; i < n; i++) {
        int i0 = (int)(std::floor(p[4*i] * inv) - (float)minb[0]), i1 = (int)(std::floor(p[4*i+1] * inv) - (float)minb[1]), i2 = (int)(std::floor(p[4*i+2] * inv) - (float)minb[2]);
        a[i] = {(unsigned)(i0 + i1 * divb[0] + i2 * divb[0] * divb[1]), (unsigned)i};
    }
    b = a;
    std::sort(a.begin(), a.end(), std::less<cpi>());          // what pcl::VoxelGrid does
    std::stable_sort(b.begin(), b.end(), std::less<cpi>());   // input order inside a voxel (oracle / product)
    size_t vox = 0, diff_vox = 0, diff_order = 0; double max_ulp = 0;
    for (size_t s = 0; s < a.size();) {
        size_t e = s; while (e < a.size() && a[e].idx == a[s].idx) e++;
        float ca[4] = {0,0,0,0}, cb[4] = {0,0,0,0};
        bool same_order = true;
        for (size_t j = s; j < e; j++) { if (a[j].cloud_point_index != b[j].cloud_point_index) same_order = false; for (int k = 0; k < 4; k++) { ca[k] += p[4*a[j].cloud_point_index+k]; cb[k] += p[4*b[j].cloud_point_index+k]; } }
        bool d = false;
        for (int k = 0; k < 4; k++) { ca[k] /= (float)(e - s); cb[k] /= (float)(e - s); if (ca[k] != cb[k]) { d = true; int32_t ia, ib; memcpy(&ia, &ca[k], 4); memcpy(&ib, &cb[k], 4); max_ulp = std::max(max_ulp, (double)std::abs(ia - ib)); } }
        vox++; diff_vox += d; diff_order += !same_order; s = e;
    }
    printf("voxels %zu, with another addend order %zu, with a different centroid %zu (%.2f %%), max ulp %g\n", vox, diff_order, diff_vox, 100.0 * diff_vox / vox, max_ulp);
}
