// stand-in for <pcl/filters/voxel_grid.h>.  PCL is not installed and pcl::VoxelGrid's own source is not in the reference tree (the
// restatement is pinned to the PCL-derived pclomp::VoxelGridCovariance the tree does hold: oracle/ref_voxelgrid_cov.cpp).  This shim gives the reference's laserMapping.cpp the SAME restatement the oracle
// uses (oracle/lio_oracle.cpp voxel_downsample, through liblio_oracle.so), so that everything around it -- the reference's own
// code -- can be compared with the oracle on equal downsampled clouds.
#pragma once
#include <cstring>
#include <memory>
#include <vector>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
extern "C" int orc_voxel_downsample(const float* in_xyzi, int n, float leaf, float* out_xyzi, int cap);
namespace pcl {
// pcl::Filter: the base the localisation nodelet holds its downsample filter through (hdl_localization_nodelet.cpp:357)
template <typename PointT>
class Filter {
   public:
    using Ptr = std::shared_ptr<Filter<PointT>>;
    virtual ~Filter() {}
    virtual void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) = 0;
    virtual void filter(PointCloud<PointT>& out) = 0;
};
template <typename PointT>
class VoxelGrid : public Filter<PointT> {
    float leaf_ = 0.f;
    typename PointCloud<PointT>::ConstPtr in_;

   public:
    void setLeafSize(float lx, float, float) { leaf_ = lx; }
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) override { in_ = c; }
    void filter(PointCloud<PointT>& out) override {
        const size_t n = in_ ? in_->points.size() : 0;
        std::vector<float> a(4 * (n ? n : 1)), b(4 * (n ? n : 1));
        for (size_t i = 0; i < n; i++) {
            a[4 * i] = in_->points[i].x; a[4 * i + 1] = in_->points[i].y; a[4 * i + 2] = in_->points[i].z; a[4 * i + 3] = in_->points[i].intensity;
        }
        const int m = n ? orc_voxel_downsample(a.data(), (int)n, leaf_, b.data(), (int)n) : 0;
        PointCloud<PointT> res;
        res.points.resize(m > 0 ? m : 0);
        for (int i = 0; i < m; i++) {
            PointT p;
            p.x = b[4 * i]; p.y = b[4 * i + 1]; p.z = b[4 * i + 2]; p.intensity = b[4 * i + 3];
            res.points[i] = p;
        }
        res.width = (uint32_t)res.points.size(); res.height = 1;
        out = res;
    }
};
}  // namespace pcl
