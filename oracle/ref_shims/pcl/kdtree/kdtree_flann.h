// stand-in for <pcl/kdtree/kdtree_flann.h> (PCL / FLANN are not installed; third-party source not in the reference tree).  Included by
// IMU_Processing.hpp (unused there) and used by the local-map harness (oracle/ref_localmap.cpp) for the key-frame position tree of
// Localization (localization.h:74): pcl::KdTreeFLANN is an EXACT search (FLANN KDTreeSingleIndex, eps = 0) --
//   radiusSearch(p, r, idx, d2): every point with squared distance < r * r (FLANN's RadiusResultSet tests `dist < radius` on squared
//     values), squared distances in f32 accumulated dimension by dimension ((dx*dx + dy*dy) + dz*dz, FLANN's L2_Simple), ASCENDING
//     (KdTreeFLANN's default sorted_ = true); returns the count
//   nearestKSearch(p, k, idx, d2): the k nearest, ascending
// This stand-in is a brute-force scan with a stable sort by (d2, index): the same sets and the same order wherever distances differ.
// Test infrastructure.
#pragma once
#include <algorithm>
#include <memory>
#include <utility>
#include <vector>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
namespace pcl {
template <typename PointT>
class KdTreeFLANN {
   public:
    using Ptr = std::shared_ptr<KdTreeFLANN<PointT>>;
    using PointCloudConstPtr = typename PointCloud<PointT>::ConstPtr;
    void setInputCloud(const PointCloudConstPtr& cloud) { input_ = cloud; }
    int radiusSearch(const PointT& q, double radius, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances, unsigned int max_nn = 0) const {
        std::vector<std::pair<float, int>> all;
        const float r2 = (float)(radius * radius);
        for (size_t i = 0; input_ && i < input_->points.size(); i++) {
            const float d2 = dist2(input_->points[i], q);
            if (d2 < r2) all.emplace_back(d2, (int)i);
        }
        std::stable_sort(all.begin(), all.end());
        if (max_nn && all.size() > max_nn) all.resize(max_nn);
        fill(all, k_indices, k_sqr_distances);
        return (int)all.size();
    }
    int nearestKSearch(const PointT& q, int k, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances) const {
        std::vector<std::pair<float, int>> all;
        for (size_t i = 0; input_ && i < input_->points.size(); i++) all.emplace_back(dist2(input_->points[i], q), (int)i);
        std::stable_sort(all.begin(), all.end());
        if ((int)all.size() > k) all.resize(k);
        fill(all, k_indices, k_sqr_distances);
        return (int)all.size();
    }

   private:
    static float dist2(const PointT& p, const PointT& q) {
        const float ex = p.x - q.x, ey = p.y - q.y, ez = p.z - q.z;
        return (ex * ex + ey * ey) + ez * ez;
    }
    static void fill(const std::vector<std::pair<float, int>>& all, std::vector<int>& idx, std::vector<float>& d2) {
        idx.resize(all.size());
        d2.resize(all.size());
        for (size_t j = 0; j < all.size(); j++) { idx[j] = all[j].second; d2[j] = all[j].first; }
    }
    PointCloudConstPtr input_;
};
}  // namespace pcl
