// stand-in for <pcl/common/common.h> (harness ref_voxelgrid_cov.cpp only): getMinMax3D as PCL 1.9.1 states it -- component-wise minimum /
// maximum over the points' (x, y, z, 1) arrays; a cloud that is not dense skips the points with a non-finite coordinate.  Plain min / max: exact.
#pragma once
#include <cfloat>
#include <cmath>
#include <string>
#include <vector>
#include <Eigen/Core>
#include <pcl/point_cloud.h>
namespace pcl {
struct PCLPointField { std::string name; uint32_t offset = 0; uint8_t datatype = 0; uint32_t count = 0; };
template <typename PointT>
inline void getMinMax3D(const PointCloud<PointT>& cloud, Eigen::Vector4f& min_pt, Eigen::Vector4f& max_pt) {
    Eigen::Array4f mn, mx;
    mn.setConstant(FLT_MAX);
    mx.setConstant(-FLT_MAX);
    for (size_t i = 0; i < cloud.points.size(); i++) {
        const PointT& p = cloud.points[i];
        if (!cloud.is_dense && (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z))) continue;
        const Eigen::Array4f pt(p.x, p.y, p.z, 1.0f);
        mn = mn.min(pt);
        mx = mx.max(pt);
    }
    min_pt = mn;
    max_pt = mx;
}
// the field-limited overload: only named on a path the harness never takes (filter_field_name_ stays empty)
template <typename PointT>
inline void getMinMax3D(const typename PointCloud<PointT>::ConstPtr& cloud, const std::string&, float, float, Eigen::Vector4f& min_pt, Eigen::Vector4f& max_pt, bool = false) {
    getMinMax3D(*cloud, min_pt, max_pt);
}
template <typename PointT>
inline int getFieldIndex(const PointCloud<PointT>&, const std::string&, std::vector<PCLPointField>&) { return -1; }
}  // namespace pcl
