// stand-in for <pcl/search/search.h> (PCL is not installed; third-party source not in the reference tree): the abstract search interface
// fast_gicp::FastGICP::calculate_covariances takes (setInputCloud / getInputCloud / nearestKSearch).  Test infrastructure.
#pragma once
#include <memory>
#include <vector>
#include <pcl/point_cloud.h>
namespace pcl {
namespace search {
template <typename PointT>
class Search {
   public:
    using PointCloudConstPtr = typename pcl::PointCloud<PointT>::ConstPtr;
    virtual ~Search() {}
    virtual void setInputCloud(const PointCloudConstPtr& cloud) { input_ = cloud; }
    PointCloudConstPtr getInputCloud() const { return input_; }
    virtual int nearestKSearch(const PointT& p, int k, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances) const = 0;

   protected:
    PointCloudConstPtr input_;
};
}  // namespace search
}  // namespace pcl
