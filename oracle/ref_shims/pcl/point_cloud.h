// Minimal stand-in for <pcl/point_cloud.h>: a vector of points with the Ptr typedef the reference headers name.
#pragma once
#include <Eigen/StdVector>
#include <memory>
#include <vector>
namespace pcl {
template <typename PointT>
struct PointCloud {
    using Ptr = std::shared_ptr<PointCloud<PointT>>;
    using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
    std::vector<PointT, Eigen::aligned_allocator<PointT>> points;
    size_t size() const { return points.size(); }
    void clear() { points.clear(); }
};
}  // namespace pcl
