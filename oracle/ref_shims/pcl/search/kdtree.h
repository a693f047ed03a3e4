// stand-in for <pcl/search/kdtree.h> (PCL / FLANN are not installed; third-party source not in the reference tree).  pcl::search::KdTree is
// an EXACT k-nearest-neighbour search (FLANN KDTreeSingleIndex, eps = 0): the k points of smallest squared distance, the distance computed
// in f32 as ((dx*dx + dy*dy) + dz*dz) (FLANN's L2_Simple accumulates the dimensions in order), ascending.  This stand-in returns exactly
// that set by a ring search over a uniform grid (after ring r every unseen point is at least r cells away); equal distances are ordered by
// point index (FLANN's own tie order is an implementation detail of its tree walk: a parity test that meets a tie at the k-th place must
// not assert on that neighbour).  Test infrastructure; also included by ndt_cuda.hpp, which uses nothing from it.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <unordered_map>
#include <utility>
#include <pcl/search/search.h>
namespace pcl {
namespace search {
template <typename PointT>
class KdTree : public Search<PointT> {
   public:
    using PointCloudConstPtr = typename Search<PointT>::PointCloudConstPtr;
    using Ptr = std::shared_ptr<KdTree<PointT>>;
    static float& cell_size() { static float c = 1.0f; return c; }  // grid cell of the stand-in (any value gives the same answers)
    void setInputCloud(const PointCloudConstPtr& cloud) override {
        this->input_ = cloud;
        cell_ = cell_size();
        grid_.clear();
        if (!cloud) return;
        for (size_t i = 0; i < cloud->points.size(); i++) {
            const PointT& p = cloud->points[i];
            grid_[key(cidx(p.x), cidx(p.y), cidx(p.z))].push_back((int)i);
        }
        lo_[0] = lo_[1] = lo_[2] = INT32_MAX;
        hi_[0] = hi_[1] = hi_[2] = INT32_MIN;
        for (size_t i = 0; i < cloud->points.size(); i++) {
            const PointT& p = cloud->points[i];
            const int c[3] = {cidx(p.x), cidx(p.y), cidx(p.z)};
            for (int a = 0; a < 3; a++) { lo_[a] = std::min(lo_[a], c[a]); hi_[a] = std::max(hi_[a], c[a]); }
        }
    }
    int nearestKSearch(const PointT& q, int k, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances) const override {
        std::vector<std::pair<float, int>> best;  // ascending (d2, index), at most k
        const int c[3] = {cidx(q.x), cidx(q.y), cidx(q.z)};
        int rmax = 0;
        for (int a = 0; a < 3; a++) rmax = std::max(rmax, std::max(std::abs(c[a] - lo_[a]), std::abs(hi_[a] - c[a])));
        for (int r = 0; r <= rmax + 1; r++) {
            for (int dz = -r; dz <= r; dz++)
                for (int dy = -r; dy <= r; dy++) {
                    const bool face = std::abs(dz) == r || std::abs(dy) == r;
                    for (int dx = -r; dx <= r; dx += (face || r == 0) ? 1 : 2 * r) {
                        auto it = grid_.find(key(c[0] + dx, c[1] + dy, c[2] + dz));
                        if (it == grid_.end()) continue;
                        for (int i : it->second) {
                            const PointT& p = this->input_->points[i];
                            const float ex = p.x - q.x, ey = p.y - q.y, ez = p.z - q.z;
                            const float d2 = (ex * ex + ey * ey) + ez * ez;
                            const std::pair<float, int> cand(d2, i);
                            if ((int)best.size() == k && !(cand < best.back())) continue;
                            best.insert(std::upper_bound(best.begin(), best.end(), cand), cand);
                            if ((int)best.size() > k) best.pop_back();
                        }
                    }
                }
            const float reach = (float)r * cell_;
            if ((int)best.size() == k && best.back().first <= reach * reach) break;
        }
        k_indices.resize(best.size());
        k_sqr_distances.resize(best.size());
        for (size_t j = 0; j < best.size(); j++) { k_indices[j] = best[j].second; k_sqr_distances[j] = best[j].first; }
        return (int)best.size();
    }

   private:
    int cidx(float v) const { return (int)std::floor(v / cell_); }
    static uint64_t key(int x, int y, int z) { return ((uint64_t)(uint32_t)(x & 0x1FFFFF) << 42) | ((uint64_t)(uint32_t)(y & 0x1FFFFF) << 21) | (uint64_t)(uint32_t)(z & 0x1FFFFF); }
    float cell_ = 1.0f;
    int lo_[3] = {0, 0, 0}, hi_[3] = {0, 0, 0};
    std::unordered_map<uint64_t, std::vector<int>> grid_;
};
}  // namespace search
}  // namespace pcl
