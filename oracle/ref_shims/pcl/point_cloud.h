// Minimal stand-in for <pcl/point_cloud.h>: a vector of points with the members the reference sources name.
#pragma once
#include <Eigen/StdVector>
#include <cstdint>
#include <memory>
#include <vector>
namespace pcl {
struct PCLHeader {
    uint32_t seq = 0;
    uint64_t stamp = 0;
    std::string frame_id;
};
template <typename PointT>
struct PointCloud {
    using Ptr = std::shared_ptr<PointCloud<PointT>>;
    using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
    using VectorType = std::vector<PointT, Eigen::aligned_allocator<PointT>>;
    using iterator = typename VectorType::iterator;
    using const_iterator = typename VectorType::const_iterator;
    PCLHeader header;
    VectorType points;
    uint32_t width = 0, height = 0;
    bool is_dense = true;
    PointCloud() = default;
    PointCloud(uint32_t w, uint32_t h) : points((size_t)w * h), width(w), height(h) {}
    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void clear() { points.clear(); width = height = 0; }
    void resize(size_t n) { points.resize(n); width = (uint32_t)n; height = 1; }
    void reserve(size_t n) { points.reserve(n); }
    void push_back(const PointT& p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
    PointT& operator[](size_t i) { return points[i]; }
    const PointT& operator[](size_t i) const { return points[i]; }
    PointT& back() { return points.back(); }
    const PointT& back() const { return points.back(); }
    PointT& at(size_t i) { return points.at(i); }
    const PointT& at(size_t i) const { return points.at(i); }
    iterator begin() { return points.begin(); }
    iterator end() { return points.end(); }
    const_iterator begin() const { return points.begin(); }
    const_iterator end() const { return points.end(); }
    PointCloud& operator+=(const PointCloud& o) {
        points.insert(points.end(), o.points.begin(), o.points.end());
        width = (uint32_t)points.size(); height = 1;
        return *this;
    }
};
}  // namespace pcl
