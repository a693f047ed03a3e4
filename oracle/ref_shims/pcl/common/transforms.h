// stand-in for <pcl/common/transforms.h>: pcl::transformPoint as PCL 1.9.1 publishes it (common/include/pcl/common/impl/transforms.hpp:
// `ret = point; ret.getVector3fMap() = transform * point.getVector3fMap();`) -- third-party source that is not in the tree
#pragma once
#include <Eigen/Geometry>
#include <pcl/point_cloud.h>
namespace pcl {
template <typename PointT, typename Scalar>
inline PointT transformPoint(const PointT& point, const Eigen::Transform<Scalar, 3, Eigen::Affine>& transform) {
    PointT ret = point;
    ret.getVector3fMap() = transform * point.getVector3fMap();
    return ret;
}
}  // namespace pcl
